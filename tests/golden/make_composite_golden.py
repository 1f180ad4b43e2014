#!/usr/bin/env python3
"""Generate tests/golden/composite_golden.npz: the reference's per-frame compositing, EXECUTED from its own source.

/root/reference/geometric/scripts/main.py:541-607 (painter's-algorithm compositing of the per-object masks / normal maps /
depth maps into frame maps + the JSON record) is a statement block in the middle of a long function that cannot run as a
whole here (chainer renderer, CUDA, datasets).  This script takes exactly those statements with `ast` from where they lie
and executes them on seeded inputs, with
  * `Transforms.to_pil_image / resize / to_tensor`: the three torchvision.transforms.functional helpers the block calls,
    stubbed from torchvision 0.2.1's published behaviour on top of the REAL Pillow of this image (torchvision is absent);
  * `.cuda()` as the identity, `FLAGS.render_size`, `dataset.Camera` (focal 725, u0 620.5, v0 187: derender3d/datasets.py:207-213).
The fixture stores the inputs and what the block left in `_image_instance_map`, `_image_normal_map`, `_image_depth_map`,
`json_obj`; tests/test_composite.py holds oracle/composite_oracle.py (the restatement the GPU tests compare the HIP kernel
with) against it bit for bit.  Runs only where /root/reference exists.
"""
import ast
import json
import os
import sys
import types

import numpy as np
import PIL.Image
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT]
REF = os.environ.get('SDN_REFERENCE_ROOT', '/root/reference')
MAIN = os.path.join(REF, 'geometric', 'scripts', 'main.py')
FIRST, LAST = 541, 607


def reference_block():
    src = open(MAIN).read()
    tree = ast.parse(src)
    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef) and fn.lineno < FIRST and fn.end_lineno >= LAST:
            body = [st for st in fn.body if FIRST <= st.lineno and st.end_lineno <= LAST]
            assert body and body[0].lineno == FIRST and body[-1].end_lineno == LAST, (body[0].lineno, body[-1].end_lineno)
            first = ast.get_source_segment(src, body[0])
            assert first.startswith('index_objs = to_numpy(torch.sort(_depths[:, 0]'), first
            assert isinstance(body[-1], ast.For)
            return compile(ast.Module(body=body, type_ignores=[]), MAIN, 'exec')
    raise RuntimeError('block not found')


def torchvision_functional_stub():
    """torchvision 0.2.1 transforms/functional.py: to_pil_image, resize, to_tensor for the cases the block produces"""
    m = types.SimpleNamespace()

    def to_pil_image(pic, mode=None):
        if isinstance(pic, torch.Tensor):
            if isinstance(pic, torch.FloatTensor) or pic.is_floating_point():
                pic = pic.mul(255).byte()
            npimg = np.transpose(pic.numpy(), (1, 2, 0))
        else:
            npimg = pic
        if npimg.shape[2] == 1:
            npimg = npimg[:, :, 0]
            mode_ = {np.dtype('uint8'): 'L', np.dtype('int16'): 'I;16', np.dtype('int32'): 'I', np.dtype('float32'): 'F'}[npimg.dtype]
            return PIL.Image.fromarray(npimg, mode=mode_)
        assert npimg.dtype == np.uint8 and npimg.shape[2] == 3
        return PIL.Image.fromarray(npimg, mode='RGB')

    def resize(img, size, interpolation=PIL.Image.BILINEAR):
        assert isinstance(size, (tuple, list)) and len(size) == 2
        return img.resize(size[::-1], interpolation)

    def to_tensor(pic):
        if pic.mode == 'I':
            img = torch.from_numpy(np.asarray(pic, np.int32))
        elif pic.mode == 'I;16':
            img = torch.from_numpy(np.asarray(pic, np.int16))
        elif pic.mode == 'F':
            img = torch.from_numpy(np.array(pic, np.float32, copy=True))
        else:
            img = torch.ByteTensor(torch.ByteStorage.from_buffer(pic.tobytes()))
        nchannel = {'YCbCr': 3, 'I;16': 1}.get(pic.mode, len(pic.mode))
        img = img.view(pic.size[1], pic.size[0], nchannel)
        img = img.transpose(0, 1).transpose(0, 2).contiguous()
        return img.float().div(255) if isinstance(img, torch.ByteTensor) else img
    m.to_pil_image, m.resize, m.to_tensor = to_pil_image, resize, to_tensor
    return m


def make_inputs(seed, n, R, height, width):
    g = torch.Generator().manual_seed(seed)
    masks = torch.zeros(n, 1, R, R)
    normals = torch.zeros(n, 3, R, R)
    depth_maps = torch.full((n, 1, R, R), 100.0)
    for i in range(n):
        y0, x0 = int(torch.randint(2, R // 3, (1,), generator=g)), int(torch.randint(2, R // 3, (1,), generator=g))
        y1, x1 = int(torch.randint(2 * R // 3, R - 2, (1,), generator=g)), int(torch.randint(2 * R // 3, R - 2, (1,), generator=g))
        soft = torch.rand(y1 - y0, x1 - x0, generator=g)
        masks[i, 0, y0:y1, x0:x1] = (soft > 0.15).float() * (0.5 + 0.5 * torch.rand(y1 - y0, x1 - x0, generator=g)).round()
        masks[i, 0, y0:y1, x0] = 0.5                     # anti-aliased edge values: round() decides
        normals[i] = torch.rand(3, R, R, generator=g) * 2 - 1
        depth_maps[i, 0, y0:y1, x0:x1] = 5 + 60 * torch.rand(y1 - y0, x1 - x0, generator=g)
    depths = 5 + 40 * torch.rand(n, 1, generator=g)
    zooms = 0.5 + 3.0 * torch.rand(n, 1, generator=g)
    center2ds = torch.stack([(torch.rand(n, generator=g) - 0.5) * 0.35, (torch.rand(n, generator=g) - 0.5) * 1.4], 1)
    interests = torch.rand(n, generator=g) > 0.2
    image_masks = (torch.rand(n, 1, height, width, generator=g) > 0.97).float()
    alphas = torch.rand(n, 1, generator=g) * 6 - 3
    class_ids = torch.randint(0, 8, (n,), generator=g)
    return dict(masks=masks, normals=normals, depth_maps=depth_maps, depths=depths, zooms=zooms, center2ds=center2ds,
                interests=interests, image_masks=image_masks, alphas=alphas, class_ids=class_ids)


def run_case(code, inp, R, height, width, operations):
    torch.Tensor.cuda = lambda self, *a, **k: self       # the block's .cuda() calls: identity on this CPU-only box
    ns = {
        'torch': torch, 'np': np, 'PIL': PIL, 'Transforms': torchvision_functional_stub(),
        'to_numpy': lambda t: t.detach().cpu().numpy(),
        'FLAGS': types.SimpleNamespace(render_size=R),
        'dataset': types.SimpleNamespace(Camera=types.SimpleNamespace(focal=725.0, u0=620.5, v0=187.0)),
        'height': height, 'width': width, 'metas': None, 'operations': operations, 'print': lambda *a, **k: None,
        '_depths': inp['depths'], '_alphas': inp['alphas'], '_zooms': inp['zooms'], '_center2ds': inp['center2ds'],
        '_masks': inp['masks'], '_normals': inp['normals'], '_depth_maps': inp['depth_maps'], 'interests': inp['interests'],
        'class_ids': inp['class_ids'], 'image_masks': inp['image_masks'],
    }
    exec(code, ns)
    return ns['_image_instance_map'], ns['_image_normal_map'], ns['_image_depth_map'], ns['json_obj'], ns['index_objs']


def main():
    code = reference_block()
    out = {}
    cases = [('vkitti', 11, 10, 96, 375, 1242, None), ('small', 12, 6, 48, 375, 1242, None), ('edit', 13, 8, 64, 375, 1242, 'x')]
    for name, seed, n, R, height, width, operations in cases:
        inp = make_inputs(seed, n, R, height, width)
        inst, nrm, dep, js, order = run_case(code, inp, R, height, width, operations)
        for k, v in inp.items():
            out['%s/%s' % (name, k)] = v.numpy()
        out['%s/instance' % name] = inst.numpy()
        out['%s/normal' % name] = nrm.numpy()
        out['%s/depth' % name] = dep.numpy()
        out['%s/order' % name] = np.asarray(order, np.int64)
        out['%s/json' % name] = np.asarray(json.dumps(js, sort_keys=True))
        out['%s/geometry' % name] = np.asarray([R, height, width, 0 if operations is None else 1], np.int64)
        print(name, 'objects', n, 'covered pixels', int((inst > 0).sum()), 'json entries', len(js))
    np.savez_compressed(os.path.join(HERE, 'composite_golden.npz'), **out)
    print('wrote', os.path.join(HERE, 'composite_golden.npz'), os.path.getsize(os.path.join(HERE, 'composite_golden.npz')))


if __name__ == '__main__':
    main()
