"""The textural input assembly (textural/data/assemble.py) with CUDA tensors against the PIL-based loader restatement
(oracle/loader_oracle.py): the same bit-exact comparison as tests/test_assemble.py, on the device."""
import os
import sys

import PIL.Image
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural'), os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [0, 2, 5])
def test_assembled_item_on_the_device_equals_the_loader(case):
    from data import assemble as asm
    from oracle import loader_oracle as lo
    from test_assemble import CASES, _frame, _opt
    opt = _opt(**CASES[case])
    segm, image, inst, normal, js = _frame(case)
    oh, ow = asm.load_size_after_scaling(opt, 375, 1242)
    params = {'crop_pos': (max(0, ow - opt.fineWidth) // 3, max(0, oh - opt.fineHeight) // 2), 'flip': True}
    ref = lo.get_item(opt, params, PIL.Image.fromarray(segm, 'L'), PIL.Image.fromarray(image, 'RGB'),
                      PIL.Image.fromarray(inst, 'L'), PIL.Image.fromarray(inst, 'L'), js, PIL.Image.fromarray(normal, 'RGB'))
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous().cuda()
    got = asm.assemble_item(opt, params, t(segm), t(image), t(inst), t(inst), js, t(normal))
    for k in ('label', 'inst', 'image', 'pose', 'normal'):
        assert got[k].is_cuda and got[k].dtype == ref[k].dtype
        assert torch.equal(got[k].cpu(), ref[k]), k


def test_every_byte_value_survives_the_device_round_trip():
    """ToTensor's k / 255 and the loader's * 255 (vkitti_dataset.py:57, 72, 104) on the device: identical bits to the CPU
    for all 256 values, and the label ids come back integral."""
    from data import assemble as asm
    k = torch.arange(256, dtype=torch.uint8)
    cpu = k.float().div(255)
    dev = asm.to_tensor_u8(k.cuda())
    assert torch.equal(dev.cpu(), cpu)
    assert torch.equal((dev * 255.0).cpu(), cpu * 255.0)
    assert torch.equal((dev * 255.0).long().cpu(), k.long())


@pytest.mark.parametrize('wrap', [False, True])
def test_depth_feature_on_the_device(wrap):
    import numpy as np
    from data import assemble as asm
    from oracle import loader_oracle as lo
    from test_assemble import CASES, _frame, _opt
    opt = _opt(feat_depth='d', **CASES[0])
    segm, image, inst, normal, js = _frame(0)
    depth = np.random.default_rng(7).integers(0, 65536, (375, 1242)).astype(np.uint16)
    params = {'crop_pos': (0, 0), 'flip': True}
    pil = PIL.Image.fromarray(depth, 'I;16') if wrap else PIL.Image.fromarray(depth.astype(np.int32), 'I')
    ref = lo.get_item(opt, params, PIL.Image.fromarray(segm, 'L'), PIL.Image.fromarray(image, 'RGB'), depth_map=pil)
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous().cuda()
    got = asm.assemble_item(opt, params, t(segm), t(image), depth=t(depth.astype(np.int32)), depth_wrap_int16=wrap)
    assert got['depth'].is_cuda and torch.equal(got['depth'].cpu(), ref['depth'])


@pytest.mark.parametrize('ci', [0, 1, 5, 8])
def test_assembly_on_the_device_against_the_reference_loader_golden(ci):
    """tests/golden/loader_golden.npz -- the `input_dict` of the reference's own CustomDataset.__getitem__ (see
    tests/test_assemble.py::test_oracle_and_product_reproduce_the_reference_loader) -- reproduced bit for bit from CUDA tensors."""
    import json
    import numpy as np
    from data import assemble as asm
    from test_assemble import GOLD, _opt
    z = np.load(GOLD)
    p = 'c%d/' % ci
    cfg = json.loads(str(z[p + 'cfg']))
    opt = _opt(**{k: v for k, v in cfg.items() if k in ('resize_or_crop', 'loadSize', 'fineWidth', 'fineHeight', 'isTrain', 'no_flip',
                                                        'n_downsample_global', 'netG', 'n_local_enhancers', 'label_nc',
                                                        'no_instance', 'feat_pose_num_bins')})
    opt.segm_precomputed_path = 'p' if cfg['segm_precomputed'] else ''
    opt.inst_precomputed_path = 'q' if cfg['inst_precomputed'] else ''
    opt.feat_pose = 'x' if cfg['pose'] else ''
    opt.feat_normal = 'x' if cfg['normal'] else ''
    opt.feat_depth = 'd' if cfg['depth'] else ''
    params = {'crop_pos': (int(z[p + 'crop_pos'][0]), int(z[p + 'crop_pos'][1])), 'flip': bool(z[p + 'flip'])}
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous().cuda()
    src = {k: z[p + 'src_' + k] for k in ('segm', 'rgb', 'instmap', 'normalmap', 'depthmap')}
    got = asm.assemble_item(opt, params, t(src['segm']), t(src['rgb']), t(src['instmap']),
                            t(src['instmap']) if cfg['pose'] else None, json.loads(str(z[p + 'json'])),
                            t(src['normalmap']) if cfg['normal'] else None,
                            depth=t(src['depthmap'].astype(np.int32)) if cfg['depth'] else None,
                            depth_wrap_int16=(str(z[p + 'depth_mode']) == 'I;16'))
    for k in ('label', 'inst', 'image', 'pose', 'normal', 'depth'):
        want = z[p + k]
        if want.shape == ():
            assert isinstance(got[k], int) and got[k] == int(want), k
            continue
        assert got[k].is_cuda and str(got[k].dtype).replace('torch.', '') == str(want.dtype), (k, got[k].dtype)
        assert np.array_equal(got[k].cpu().numpy(), want), '%s differs in %d elements' % (k, int((got[k].cpu().numpy() != want).sum()))
