"""CPU tests of the textural input assembly (3d-sdn_amd/textural/data/assemble.py) against the PIL-based restatement of
the reference loader (oracle/loader_oracle.py): PIL-exact resizing and every option branch of
textural/data/vkitti_dataset.py:44-142 used by the 3D-SDN configurations."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import PIL.Image
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from data import assemble as asm  # noqa: E402
from oracle import loader_oracle as lo  # noqa: E402

PIL_METHOD = {'bicubic': PIL.Image.BICUBIC, 'bilinear': PIL.Image.BILINEAR, 'nearest': PIL.Image.NEAREST}


@pytest.mark.parametrize('method', ['bicubic', 'bilinear', 'nearest'])
@pytest.mark.parametrize('shape', [(375, 1242, 192, 624), (375, 1242, 368, 1248), (375, 1242, 800, 800), (100, 77, 33, 200),
                                   (64, 64, 64, 100), (50, 60, 50, 60)])
def test_resize_u8_is_bit_identical_to_pil(method, shape):
    H, W, oh, ow = shape
    rng = np.random.default_rng(H + ow)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.array(PIL.Image.fromarray(img, mode='RGB').resize((ow, oh), PIL_METHOD[method]))
    got = asm.resize_u8(torch.from_numpy(img).permute(2, 0, 1).contiguous(), oh, ow, method)
    assert np.array_equal(got.permute(1, 2, 0).numpy(), ref)


def _opt(**kw):
    o = dict(resize_or_crop='scale_width_and_crop', loadSize=624, fineWidth=624, fineHeight=192, isTrain=True, no_flip=False,
             n_downsample_global=4, netG='global', n_local_enhancers=1, label_nc=14, no_instance=False,
             segm_precomputed_path='', inst_precomputed_path='', feat_pose='x', feat_pose_num_bins=24, feat_normal='x')
    o.update(kw)
    return SimpleNamespace(**o)


def _frame(seed, H=375, W=1242):
    rng = np.random.default_rng(seed)
    segm = rng.integers(0, 14, (H, W), dtype=np.uint8)
    image = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    normal = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    inst = np.zeros((H, W), dtype=np.uint8)
    js = {}
    for k in range(1, 9):
        y0, x0 = int(rng.integers(0, H - 60)), int(rng.integers(0, W - 200))
        inst[y0:y0 + int(rng.integers(20, 60)), x0:x0 + int(rng.integers(40, 200))] = k
        if k != 5:   # one instance without a pose record
            js[str(k)] = {'class_id': 1, 'depth': 10.0, 'alpha': float(rng.uniform(-np.pi, np.pi))}
    js['77'] = {'class_id': 1, 'depth': 1.0, 'alpha': 0.3}   # a record whose instance is not in the map
    return segm, image, inst, normal, js


CASES = [
    dict(),                                                              # the training default of the reference
    dict(isTrain=False),                                                 # test time: no flip
    dict(resize_or_crop='none'),                                         # make_power_2 (375 x 1242 -> 368 x 1248)
    dict(resize_or_crop='resize_and_crop', loadSize=256, fineWidth=200, fineHeight=160),
    dict(resize_or_crop='scale_width', loadSize=800),
    dict(segm_precomputed_path='p', inst_precomputed_path='q'),          # geometric-branch outputs as inputs
    dict(feat_pose_num_bins=0),                                          # (cos, sin) pose features
    dict(feat_pose='', feat_normal=''),
]


@pytest.mark.parametrize('case', range(len(CASES)))
@pytest.mark.parametrize('flip', [False, True])
def test_assembled_item_equals_the_loader(case, flip):
    opt = _opt(**CASES[case])
    segm, image, inst, normal, js = _frame(case)
    oh, ow = asm.load_size_after_scaling(opt, 375, 1242)
    params = {'crop_pos': (max(0, ow - opt.fineWidth) // 3, max(0, oh - opt.fineHeight) // 2), 'flip': flip}
    ref = lo.get_item(opt, params, PIL.Image.fromarray(segm, 'L'), PIL.Image.fromarray(image, 'RGB'),
                      PIL.Image.fromarray(inst, 'L'), PIL.Image.fromarray(inst, 'L'), js, PIL.Image.fromarray(normal, 'RGB'))
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous()
    got = asm.assemble_item(opt, params, t(segm), t(image), t(inst), t(inst), js, t(normal))
    for k in ('label', 'inst', 'image', 'pose', 'normal'):
        if isinstance(ref[k], int):
            assert isinstance(got[k], int) and got[k] == ref[k], k
            continue
        assert got[k].dtype == ref[k].dtype and tuple(got[k].shape) == tuple(ref[k].shape), (k, got[k].dtype, ref[k].dtype)
        assert torch.equal(got[k], ref[k]), '%s differs in %d elements' % (k, int((got[k] != ref[k]).sum()))


def test_missing_instance_and_pose_files():
    opt = _opt()
    segm, image, inst, normal, js = _frame(3)
    params = {'crop_pos': (0, 0), 'flip': False}
    ref = lo.get_item(opt, params, PIL.Image.fromarray(segm, 'L'), PIL.Image.fromarray(image, 'RGB'))
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous()
    got = asm.assemble_item(opt, params, t(segm), t(image))
    assert torch.equal(got['inst'], ref['inst']) and torch.equal(got['label'], ref['label'])
    assert torch.equal(got['pose'], ref['pose']) and torch.equal(got['normal'], ref['normal'])


@pytest.mark.parametrize('mode', ['I', 'I;16'])
@pytest.mark.parametrize('case', [0, 2, 3])
def test_depth_feature_equals_the_loader(case, mode):
    """--feat_depth (vkitti_dataset.py:131-137): 16-bit depth PNG -> NEAREST geometry -> 1 - d / 65535, for both pixel
    modes PIL has used for 16-bit PNGs (torchvision 0.2.x reads 'I;16' through int16)."""
    opt = _opt(feat_depth='d', **CASES[case])
    segm, image, inst, normal, js = _frame(case)
    rng = np.random.default_rng(100 + case)
    depth = rng.integers(0, 65536, (375, 1242)).astype(np.uint16)
    depth[:4, :4] = np.array([0, 1, 32767, 32768, 65535, 65534, 255, 256, 40000, 1000, 2, 3, 4, 5, 6, 7], np.uint16).reshape(4, 4)
    oh, ow = asm.load_size_after_scaling(opt, 375, 1242)
    params = {'crop_pos': (max(0, ow - opt.fineWidth) // 3, max(0, oh - opt.fineHeight) // 2), 'flip': True}
    pil = PIL.Image.fromarray(depth.astype(np.int32), 'I') if mode == 'I' else PIL.Image.fromarray(depth, 'I;16')
    ref = lo.get_item(opt, params, PIL.Image.fromarray(segm, 'L'), PIL.Image.fromarray(image, 'RGB'), depth_map=pil)
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous()
    got = asm.assemble_item(opt, params, t(segm), t(image), depth=t(depth.astype(np.int32)), depth_wrap_int16=(mode == 'I;16'))
    assert got['depth'].dtype == ref['depth'].dtype and got['depth'].shape == ref['depth'].shape
    assert torch.equal(got['depth'], ref['depth'])
    none = asm.assemble_item(opt, params, t(segm), t(image))
    assert torch.equal(none['depth'], torch.zeros_like(none['label']))


GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'loader_golden.npz')


def _gold_cases():
    import json
    z = np.load(GOLD)
    return z, int(z['ncases']), json


@pytest.mark.parametrize('ci', range(10))
def test_oracle_and_product_reproduce_the_reference_loader(ci):
    """tests/golden/loader_golden.npz: `input_dict` of the REFERENCE's own CustomDataset.__getitem__ (textural/data/
    vkitti_dataset.py:44-142 on base_dataset.py:21-110, imported as it lies by tests/golden/make_loader_golden.py and run on a
    temporary VKITTI-shaped tree), with the crop / flip it drew.  Both the restatement (oracle/loader_oracle.py) and the
    product's tensor-op assembly (textural/data/assemble.py) must reproduce every tensor bit for bit from the same frame."""
    z, n, json = _gold_cases()
    assert n == 10
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
    segm, image, inst, normal, depth = (z[p + 'src_' + k] for k in ('segm', 'rgb', 'instmap', 'normalmap', 'depthmap'))
    js = json.loads(str(z[p + 'json']))
    depth_mode = str(z[p + 'depth_mode'])
    # ---- the restatement
    pil_depth = None
    if cfg['depth']:
        assert depth_mode in ('I;16', 'I')
        pil_depth = PIL.Image.fromarray(depth, 'I;16') if depth_mode == 'I;16' else PIL.Image.fromarray(depth.astype(np.int32), 'I')
    ref = lo.get_item(opt, params, PIL.Image.fromarray(segm, 'L'), PIL.Image.fromarray(image, 'RGB'),
                      PIL.Image.fromarray(inst, 'L'), PIL.Image.fromarray(inst, 'L') if cfg['pose'] else None, js,
                      PIL.Image.fromarray(normal, 'RGB') if cfg['normal'] else None, depth_map=pil_depth)
    # ---- the product
    t = lambda a: torch.from_numpy(a if a.ndim == 3 else a[:, :, None]).permute(2, 0, 1).contiguous()
    got = asm.assemble_item(opt, params, t(segm), t(image), t(inst), t(inst) if cfg['pose'] else None, js,
                            t(normal) if cfg['normal'] else None,
                            depth=t(depth.astype(np.int32)) if cfg['depth'] else None, depth_wrap_int16=(depth_mode == 'I;16'))
    for k in ('label', 'inst', 'image', 'pose', 'normal', 'depth'):
        want = z[p + k]
        if k == 'inst' and cfg['no_instance']:
            assert want.shape == () and int(want) == 0     # the loader leaves the integer 0 (vkitti_dataset.py:55)
            continue
        for who, have in (('oracle', ref[k]), ('product', got[k])):
            if want.shape == ():
                assert isinstance(have, int) and have == int(want), (who, k)
                continue
            assert isinstance(have, torch.Tensor), (who, k, type(have))
            assert tuple(have.shape) == want.shape and str(have.dtype).replace('torch.', '') == str(want.dtype), (who, k, have.dtype, want.dtype)
            assert np.array_equal(have.numpy(), want), '%s: %s differs in %d elements' % (who, k, int((have.numpy() != want).sum()))


def test_resize_tables_are_kept_per_device(monkeypatch):
    """r05: the index / weight tables of the resizes are uploaded once per device (`.to(device)` of a pageable host tensor is a
    synchronous copy: 14 of them per frame paced configs[4]); SDN_ASSEMBLE_UPLOAD_TABLES=1 is the A/B switch back."""
    from data import assemble as asm
    dev = torch.device('cpu')
    t1 = asm._on(dev, asm._nearest_table, 375, 368)
    assert asm._on(dev, asm._nearest_table, 375, 368) is t1 and torch.equal(t1, asm._nearest_table(375, 368))
    i1, k1 = asm._on(dev, asm._resample_table, 1242, 1248, 'bicubic')
    i2, k2 = asm._on(dev, asm._resample_table, 1242, 1248, 'bicubic')
    assert i1 is i2 and k1 is k2
    assert asm._on(dev, asm._nearest_table, 375, 184) is not t1
    img = torch.arange(3 * 10 * 12, dtype=torch.uint8).reshape(3, 10, 12)
    a = asm.resize_u8(img, 7, 9, 'bicubic')
    monkeypatch.setenv('SDN_ASSEMBLE_UPLOAD_TABLES', '1')
    assert torch.equal(asm.resize_u8(img, 7, 9, 'bicubic'), a) and torch.equal(asm.resize_nearest(img, 7, 9), asm.resize_nearest(img, 7, 9))
