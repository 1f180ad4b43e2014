"""CPU tests of the compositing host logic (3d-sdn_amd/geometric/derender3d/compositing.py): the restated Pillow
resampling tables against the real PIL of this image, the paste geometry against the reference formulas
(geometric/scripts/main.py:556-569), and the wire-format writer (main.py:604-622)."""
import json
import os
import sys

import numpy as np
import PIL.Image
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from derender3d import compositing as comp  # noqa: E402
from oracle import composite_oracle as co  # noqa: E402


@pytest.mark.parametrize('R,S', [(64, 37), (64, 100), (64, 63), (64, 65), (96, 17), (384, 225), (384, 640), (384, 383),
                                 (50, 7), (384, 384), (16, 1)])
def test_resampling_tables_reproduce_pil(R, S):
    rng = np.random.default_rng(R * 1000 + S)
    img = rng.integers(0, 256, (R, R), dtype=np.uint8)
    ref = np.array(PIL.Image.fromarray(img, mode='L').resize((S, S), PIL.Image.BILINEAR))
    assert np.array_equal(comp.resample_u8_numpy(img, S), ref)
    f = (rng.random((R, R), dtype=np.float32) * 3).astype(np.float32)
    reff = np.array(PIL.Image.fromarray(f, mode='F').resize((S, S), PIL.Image.BILINEAR))
    assert np.array_equal(comp.resample_f32_numpy(f, S), reff)


def test_paste_geometry_matches_reference_arithmetic():
    torch.manual_seed(3)
    zooms = torch.rand(50) * 3 + 0.3
    c2d = (torch.rand(50, 2) - 0.5) * 0.8
    geo = comp.paste_geometry(zooms.numpy(), c2d.numpy(), 725.0, 620.5, 187.0, 384)
    for i in range(50):
        size = int(384 / zooms[i])  # float32 tensor arithmetic + int(), main.py:560-569
        left = int(c2d[i, 1] * 725.0 + 620.5 - size // 2)
        top = int(c2d[i, 0] * 725.0 + 187.0 - size // 2)
        assert geo[i] == (size, left, top)


def test_oracle_composite_painter_order_and_defaults():
    n, R = 2, 32
    masks = torch.zeros(n, 1, R, R)
    masks[:, :, 8:24, 8:24] = 1
    normals = torch.zeros(n, 3, R, R)
    normals[0, 0] = 1.0
    normals[1, 1] = 1.0
    dm = torch.full((n, 1, R, R), 50.0)
    depths = torch.tensor([[5.0], [9.0]])       # object 0 is nearer: it must win where both cover
    zooms = torch.ones(n)
    c2d = torch.zeros(n, 2)
    inst, nrm, dep, order = co.composite_frame(masks, normals, dm, depths, zooms, c2d, torch.ones(n, dtype=torch.bool),
                                               10.0, 40.0, 30.0, 60, 80, R)
    assert order == [1, 0]
    assert set(inst.unique().tolist()) == {0.0, 1.0}           # object 1 is completely hidden behind object 0
    assert float(nrm[:, 0, 0].sub(0.5).abs().max()) == 0 and float(dep[0, 0, 0]) == 1.0
    cy, cx = 30, 40
    assert inst[0, cy, cx] == 1 and abs(float(nrm[0, cy, cx]) - 1.0) < 1e-6 and abs(float(dep[0, cy, cx]) - 0.5) < 1e-6


def test_write_frame_wire_format(tmp_path):
    H, W = 12, 20
    inst = torch.zeros(1, H, W)
    inst[0, 2:5, 3:9] = 7
    nrm = torch.rand(3, H, W)
    dep = torch.rand(1, H, W)
    js = comp.frame_json([1, 0], [True, False], [3, 4], [12.5, 3.0], [0.25, -1.0], [{'tid': 9}, {'tid': 1}])
    assert js == {1: {'class_id': 3, 'depth': 12.5, 'alpha': 0.25, 'tid': 9}}   # only interesting objects, key = index + 1
    comp.write_frame(str(tmp_path), '00001', inst, nrm, dep, js)
    assert np.array_equal(np.array(PIL.Image.open(tmp_path / '00001.png')), inst[0].numpy().astype(np.uint8))
    assert np.array_equal(np.array(PIL.Image.open(tmp_path / '00001-normal.png')),
                          nrm.mul(255).byte().numpy().transpose(1, 2, 0))
    d16 = np.array(PIL.Image.open(tmp_path / '00001-depth.png'))
    assert np.array_equal(d16.astype(np.uint16), np.uint16(dep[0].numpy() * 65535))
    assert json.load(open(tmp_path / '00001.json')) == {'1': {'class_id': 3, 'depth': 12.5, 'alpha': 0.25, 'tid': 9}}


def test_cpu_tensors_are_refused():
    z = torch.zeros(1, 1, 8, 8)
    with pytest.raises(NotImplementedError):
        comp.composite_frame(z, torch.zeros(1, 3, 8, 8), z, torch.ones(1, 1), torch.ones(1), torch.zeros(1, 2),
                             torch.ones(1), 1.0, 4.0, 4.0, 8, 8)


GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'composite_golden.npz')


def _gold_case(z, name):
    t = {k: torch.from_numpy(z['%s/%s' % (name, k)]) for k in ('masks', 'normals', 'depth_maps', 'depths', 'zooms', 'center2ds',
                                                                'interests', 'image_masks', 'alphas', 'class_ids')}
    R, H, W, edit = (int(v) for v in z['%s/geometry' % name])
    return t, R, H, W, bool(edit)


@pytest.mark.parametrize('name', ['vkitti', 'small', 'edit'])
def test_oracle_reproduces_the_reference_block(name):
    """tests/golden/composite_golden.npz holds what the reference's OWN statements (geometric/scripts/main.py:541-607, taken
    with ast and executed by tests/golden/make_composite_golden.py) left in the frame maps and the JSON record; the
    restatement the GPU tests compare the HIP kernel with must reproduce it bit for bit.  'edit': `operations` is set, so
    uninteresting objects do not stamp their Mask R-CNN masks (main.py:598-601)."""
    z = np.load(GOLD)
    t, R, H, W, edit = _gold_case(z, name)
    inst, nrm, dep, order = co.composite_frame(t['masks'], t['normals'], t['depth_maps'], t['depths'], t['zooms'], t['center2ds'],
                                               t['interests'], 725.0, 620.5, 187.0, H, W, R,
                                               image_masks=None if edit else t['image_masks'])
    assert order == z['%s/order' % name].tolist()
    assert np.array_equal(inst.numpy(), z['%s/instance' % name])
    assert np.array_equal(nrm.numpy(), z['%s/normal' % name])
    assert np.array_equal(dep.numpy(), z['%s/depth' % name])
    js = comp.frame_json(order, t['interests'].tolist(), t['class_ids'].tolist(), t['depths'][:, 0].tolist(),
                         t['alphas'][:, 0].tolist())
    ref = json.loads(str(z['%s/json' % name]))
    assert {str(k): v for k, v in js.items()} == ref
