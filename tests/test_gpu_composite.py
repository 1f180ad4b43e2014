"""GPU parity of the device compositing (sdn_composite_frame through derender3d/compositing.py) against the PIL-based
restatement of geometric/scripts/main.py:541-602 (oracle/composite_oracle.py): bit-identical instance / normal / depth
maps, including identity-size pastes, up- and down-scaling, objects hanging over the frame border and occlusion."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu


def _objects(n, R, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, 1, R // 8, R // 8, generator=g)
    masks = torch.nn.functional.interpolate((base > 0.45).float(), size=(R, R), mode='bilinear', align_corners=False)
    normals = torch.nn.functional.normalize(torch.randn(n, 3, R, R, generator=g), dim=1) * masks
    depth_maps = torch.rand(n, 1, R, R, generator=g) * 80 + 1
    depths = torch.rand(n, 1, generator=g) * 40 + 3
    return masks, normals, depth_maps, depths


@pytest.mark.parametrize('case', ['vkitti16', 'small_mixed'])
def test_device_compositing_is_bit_identical_to_the_pil_path(case):
    from derender3d import compositing as comp
    from oracle import composite_oracle as co
    if case == 'vkitti16':
        n, R, H, W, focal, u0, v0 = 16, 384, 375, 1242, 725.0, 620.5, 187.0
        g = torch.Generator().manual_seed(5)
        zooms = torch.rand(n, generator=g) * 3 + 0.5
        zooms[0] = 1.0                                   # identity-size paste
        c2d = torch.stack([(torch.rand(n, generator=g) - 0.5) * 0.5, (torch.rand(n, generator=g) - 0.5) * 1.9], 1)
    else:
        n, R, H, W, focal, u0, v0 = 5, 64, 60, 90, 40.0, 45.0, 30.0
        zooms = torch.tensor([1.0, 0.55, 2.3, 7.9, 1.31])
        c2d = torch.tensor([[0.0, 0.0], [0.6, -1.0], [-0.7, 1.0], [0.1, 0.2], [0.2, 0.4]])
    masks, normals, depth_maps, depths = _objects(n, R, 11)
    interests = torch.ones(n, dtype=torch.bool)
    ref = co.composite_frame(masks, normals, depth_maps, depths, zooms, c2d, interests, focal, u0, v0, H, W, R)
    got = comp.composite_frame(masks.cuda(), normals.cuda(), depth_maps.cuda(), depths.cuda(), zooms.cuda(), c2d.cuda(),
                               interests.cuda(), focal, u0, v0, H, W, R)
    assert got[3] == ref[3]
    for name, a, b in zip(('instance', 'normal', 'depth'), got[:3], ref[:3]):
        assert torch.equal(a.cpu(), b), '%s map differs in %d pixels' % (name, int((a.cpu() != b).sum()))
    assert len(ref[0].unique()) > 3   # several objects visible


def test_uninteresting_objects_use_the_given_frame_masks():
    from derender3d import compositing as comp
    from oracle import composite_oracle as co
    n, R, H, W = 3, 32, 40, 50
    masks, normals, depth_maps, depths = _objects(n, R, 2)
    zooms, c2d = torch.tensor([1.0, 1.6, 0.8]), torch.tensor([[0.0, 0.0], [0.1, 0.3], [-0.2, -0.2]])
    interests = torch.tensor([True, False, True])
    image_masks = (torch.rand(n, 1, H, W) > 0.7).float()
    ref = co.composite_frame(masks, normals, depth_maps, depths, zooms, c2d, interests, 20.0, 25.0, 20.0, H, W, R, image_masks)
    got = comp.composite_frame(masks.cuda(), normals.cuda(), depth_maps.cuda(), depths.cuda(), zooms.cuda(), c2d.cuda(),
                               interests.cuda(), 20.0, 25.0, 20.0, H, W, R, image_masks.cuda())
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a.cpu(), b)


@pytest.mark.parametrize('name', ['vkitti', 'small', 'edit'])
def test_device_compositing_against_the_reference_block(name):
    """sdn_composite_frame against tests/golden/composite_golden.npz: the frame maps the reference's own statements
    (geometric/scripts/main.py:541-607, executed by tests/golden/make_composite_golden.py) produced -- bit-identical."""
    import numpy as np
    from derender3d import compositing as comp
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'composite_golden.npz'))
    t = {k: torch.from_numpy(z['%s/%s' % (name, k)]).cuda() for k in ('masks', 'normals', 'depth_maps', 'depths', 'zooms',
                                                                       'center2ds', 'interests', 'image_masks')}
    R, H, W, edit = (int(v) for v in z['%s/geometry' % name])
    got = comp.composite_frame(t['masks'], t['normals'], t['depth_maps'], t['depths'], t['zooms'], t['center2ds'], t['interests'],
                               725.0, 620.5, 187.0, H, W, R, image_masks=None if edit else t['image_masks'])
    assert got[3] == z['%s/order' % name].tolist()
    for key, a in zip(('instance', 'normal', 'depth'), got[:3]):
        b = torch.from_numpy(z['%s/%s' % (name, key)])
        assert torch.equal(a.cpu(), b), '%s map differs in %d pixels' % (key, int((a.cpu() != b).sum()))


def test_host_state_path_equals_the_per_tensor_reads():
    """compositing.host_state (r05): the scalars compositing needs on the host, for several frames, in ONE device-to-host copy;
    composite_frame(host=...) then reads nothing back -- same painter order, bit-identical maps."""
    from derender3d import compositing as comp
    n, R, H, W, focal, u0, v0 = 10, 96, 120, 300, 200.0, 150.0, 60.0
    frames = []
    for f in range(3):
        masks, normals, depth_maps, depths = _objects(n, R, 40 + f)
        g = torch.Generator().manual_seed(90 + f)
        zooms = torch.rand(n, 1, generator=g) * 2 + 0.5
        c2d = torch.stack([(torch.rand(n, generator=g) - 0.5) * 0.4, (torch.rand(n, generator=g) - 0.5) * 1.2], 1)
        alphas = torch.rand(n, 1, generator=g)
        frames.append([t.cuda() for t in (masks, normals, depth_maps, depths, zooms, c2d, alphas)])
    st = lambda k: torch.stack([fr[k] for fr in frames])   # noqa: E731
    host = comp.host_state(st(3), st(4), st(5), st(6))
    assert host.shape == (3, n, 5)
    interests = torch.ones(n, dtype=torch.bool)
    for f, (masks, normals, depth_maps, depths, zooms, c2d, alphas) in enumerate(frames):
        a = comp.composite_frame(masks, normals, depth_maps, depths, zooms, c2d, interests.cuda(), focal, u0, v0, H, W, R)
        b = comp.composite_frame(masks, normals, depth_maps, depths, zooms, c2d, [True] * n, focal, u0, v0, H, W, R, host=host[f])
        assert a[3] == b[3]
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        assert host[f][:, 4].tolist() == alphas[:, 0].tolist() and host[f][:, 0].tolist() == depths[:, 0].tolist()
