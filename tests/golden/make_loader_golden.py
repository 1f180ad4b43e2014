#!/usr/bin/env python3
"""Generate tests/golden/loader_golden.npz: the reference's OWN data loader, executed.

/root/reference/textural/data/vkitti_dataset.py (CustomDataset.__getitem__, :44-142) on top of data/base_dataset.py (:21-110) is
imported as it lies -- with `torchvision.transforms` stubbed (Compose / Lambda / Scale / ToTensor / Normalize from
torchvision 0.2.1's published behaviour on the REAL Pillow of this image; torchvision itself is absent) -- pointed at a
temporary VKITTI-shaped directory tree holding one synthetic frame (label map, rgb, instance map, the geometric branch's wire
files NNNNN.png / .json / -normal.png / -depth.png), and asked for that item under the option sets the 3D-SDN configurations
use.  The fixture stores the frame's source arrays, the random crop / flip the loader drew, and every tensor of the returned
`input_dict`.  tests/test_assemble.py holds oracle/loader_oracle.py (the restatement) AND the product's
textural/data/assemble.py against it, bit for bit.  Runs only where /root/reference exists.
"""
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import PIL.Image
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('SDN_REFERENCE_ROOT', '/root/reference')


def torchvision_stub():
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')

    class Compose(object):
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, img):
            for t in self.transforms:
                img = t(img)
            return img

    class Lambda(object):
        def __init__(self, lambd):
            self.lambd = lambd

        def __call__(self, img):
            return self.lambd(img)

    class Scale(object):   # torchvision 0.2.1 Resize with a (h, w) size
        def __init__(self, size, interpolation=PIL.Image.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            return img.resize(tuple(self.size[::-1]), self.interpolation)

    class ToTensor(object):
        def __call__(self, pic):
            if pic.mode == 'I':
                img = torch.from_numpy(np.asarray(pic, np.int32))
            elif pic.mode == 'I;16':
                img = torch.from_numpy(np.asarray(pic, np.int16))
            elif pic.mode == 'F':
                img = torch.from_numpy(np.asarray(pic, np.float32))
            else:
                img = torch.ByteTensor(torch.ByteStorage.from_buffer(pic.tobytes()))
            nchannel = {'YCbCr': 3, 'I;16': 1}.get(pic.mode, len(pic.mode))
            img = img.view(pic.size[1], pic.size[0], nchannel)
            img = img.transpose(0, 1).transpose(0, 2).contiguous()
            return img.float().div(255) if isinstance(img, torch.ByteTensor) else img

    class Normalize(object):
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, tensor):
            for t, m, s in zip(tensor, self.mean, self.std):
                t.sub_(m).div_(s)
            return tensor

    class ColorJitter(object):
        def __init__(self, *a, **k):
            pass
    tr.Compose, tr.Lambda, tr.Scale, tr.ToTensor, tr.Normalize, tr.ColorJitter = Compose, Lambda, Scale, ToTensor, Normalize, ColorJitter
    tv.transforms = tr
    return tv, tr


def _textured(rng, H, W, C, top):
    """smooth ramps + blocks + 8 % salt noise: exercises the resampling filters (edges, clamping at 0 / top) and still
    compresses (a pure-noise fixture would be 50 MB)"""
    y, x = np.mgrid[0:H, 0:W]
    a = np.stack([(np.sin(x / (7.0 + 3 * c)) * np.cos(y / (5.0 + 2 * c)) * 0.5 + 0.5) * top for c in range(C)], -1)
    for _ in range(12):
        y0, x0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
        a[y0:y0 + int(rng.integers(4, H // 3)), x0:x0 + int(rng.integers(4, W // 3))] = rng.integers(0, top + 1, C)
    noise = rng.random((H, W)) < 0.08
    a[noise] = rng.integers(0, top + 1, (int(noise.sum()), C))
    return a


def frame(seed, H=375, W=1242):
    rng = np.random.default_rng(seed)
    segm = (_textured(rng, H, W, 1, 13)[:, :, 0]).astype(np.uint8)
    image = _textured(rng, H, W, 3, 255).astype(np.uint8)
    normal = _textured(rng, H, W, 3, 255).astype(np.uint8)
    depth = _textured(rng, H, W, 1, 65535)[:, :, 0].astype(np.uint16)
    inst = np.zeros((H, W), dtype=np.uint8)
    js = {}
    for k in range(1, 9):
        y0, x0 = int(rng.integers(0, H - H // 6)), int(rng.integers(0, W - W // 6))
        inst[y0:y0 + int(rng.integers(H // 18, H // 6)), x0:x0 + int(rng.integers(W // 30, W // 6))] = k
        if k != 5:
            js[str(k)] = {'class_id': 1, 'depth': 10.0, 'alpha': float(rng.uniform(-np.pi, np.pi))}
    js['77'] = {'class_id': 1, 'depth': 1.0, 'alpha': 0.3}
    return segm, image, inst, normal, depth, js


BASE = dict(resize_or_crop='scale_width_and_crop', loadSize=624, fineWidth=624, fineHeight=192, isTrain=True, no_flip=False,
            n_downsample_global=4, netG='global', n_local_enhancers=1, label_nc=14, no_instance=False, use_augmentation=False,
            load_features=False, segm_precomputed=False, inst_precomputed=False, pose=True, feat_pose_num_bins=24, normal=True,
            depth=False)
SMALL = dict(frame=(94, 311), loadSize=156, fineWidth=156, fineHeight=48)   # a quarter-size frame for the option branches
CASES = [
    dict(),                                                       # the reference's training default at the VKITTI size
    dict(frame=(94, 311), resize_or_crop='none'),                 # make_power_2: 94 x 311 -> 96 x 304 (16-pixel grid)
    dict(SMALL, isTrain=False),
    dict(SMALL, resize_or_crop='resize_and_crop', loadSize=96, fineWidth=80, fineHeight=64),
    dict(SMALL, resize_or_crop='scale_width', loadSize=200),
    dict(SMALL, segm_precomputed=True, inst_precomputed=True),    # the geometric branch's outputs as inputs
    dict(SMALL, feat_pose_num_bins=0),
    dict(SMALL, pose=False, normal=False),
    dict(SMALL, depth=True, isTrain=False),
    dict(SMALL, no_instance=True),
]


def main():
    tv, tr = torchvision_stub()
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tr
    sys.path.insert(0, os.path.join(REF, 'textural'))
    from data import vkitti_dataset as vd          # the reference module, as it lies
    out = {'ncases': np.int64(len(CASES))}
    for ci, over in enumerate(CASES):
        cfg = dict(BASE)
        cfg.update(over)
        tmp = tempfile.mkdtemp(prefix='loader_golden_')
        try:
            segm, image, inst, normal, depth, js = frame(ci, *cfg.get('frame', (375, 1242)))
            o = types.SimpleNamespace(**{k: v for k, v in cfg.items() if k not in ('segm_precomputed', 'inst_precomputed', 'pose',
                                                                                    'normal', 'depth', 'frame')})
            o.dataroot = os.path.join(tmp, 'root')
            o.segm_precomputed_path = os.path.join(tmp, 'segm') if cfg['segm_precomputed'] else ''
            o.inst_precomputed_path = os.path.join(tmp, 'inst') if cfg['inst_precomputed'] else ''
            o.feat_pose = os.path.join(tmp, 'geo') if cfg['pose'] else ''
            o.feat_normal = os.path.join(tmp, 'geo') if cfg['normal'] else ''
            o.feat_depth = os.path.join(tmp, 'geo') if cfg['depth'] else ''
            random.seed(1000 + ci)
            ds = vd.CustomDataset()
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                ds.initialize(o)
            rel = ds.list[0]

            def put(root, name, arr, mode):
                path = os.path.join(root, name)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                PIL.Image.fromarray(arr, mode).save(path)
            put(ds.root_segm, rel, segm, 'L')
            put(ds.root_img, rel, image, 'RGB')
            put(ds.root_inst, rel, inst, 'L')
            if cfg['pose']:
                put(o.feat_pose, rel, inst, 'L')
                with open(os.path.join(o.feat_pose, rel).replace('png', 'json'), 'w') as f:
                    json.dump(js, f)
            if cfg['normal']:
                put(o.feat_normal, rel.replace('.png', '-normal.png'), normal, 'RGB')
            if cfg['depth']:
                put(o.feat_depth, rel.replace('.png', '-depth.png'), depth, 'I;16')
            drawn = {}
            real = vd.get_params

            def spy(opt, size):
                p = real(opt, size)
                drawn.update(p)
                return p
            vd.get_params = spy
            try:
                item = ds[0]
            finally:
                vd.get_params = real
            p = 'c%d/' % ci
            out[p + 'cfg'] = np.asarray(json.dumps(cfg, sort_keys=True))
            out[p + 'crop_pos'] = np.asarray([int(drawn['crop_pos'][0]), int(drawn['crop_pos'][1])], np.int64)
            out[p + 'flip'] = np.asarray(bool(drawn['flip']))
            out[p + 'depth_mode'] = np.asarray(PIL.Image.open(os.path.join(o.feat_depth, rel.replace('.png', '-depth.png'))).mode
                                               if cfg['depth'] else '')
            for k, a in (('segm', segm), ('rgb', image), ('instmap', inst), ('normalmap', normal), ('depthmap', depth)):
                out[p + 'src_' + k] = a
            out[p + 'json'] = np.asarray(json.dumps(js, sort_keys=True))
            for k in ('label', 'inst', 'image', 'pose', 'normal', 'depth'):
                v = item[k]
                out[p + k] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            print(ci, over, {k: (tuple(item[k].shape), str(item[k].dtype)) if isinstance(item[k], torch.Tensor) else item[k]
                             for k in ('label', 'inst', 'image', 'pose', 'normal', 'depth')}, drawn)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    path = os.path.join(HERE, 'loader_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
