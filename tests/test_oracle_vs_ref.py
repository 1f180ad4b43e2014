"""The C restatement (oracle/raster_oracle.c) must equal, bit for bit, the reference's own kernel strings
compiled for the CPU (oracle/_ref, built by oracle/build_ref.py where /root/reference exists)."""
import numpy as np
import pytest

from oracle import raster_np as rn
from util import biteq, random_soup

pytestmark = pytest.mark.skipif(not rn.have_ref(), reason='oracle/_ref not built (needs /root/reference once)')

MAPS = ['face_index_map', 'weight_map', 'depth_map', 'face_inv_map', 'rgb_map', 'alpha_map', 'sampling_index_map',
        'sampling_weight_map']


@pytest.mark.parametrize('nf,is_,scale,ts', [(50, 32, 0.3, 2), (300, 48, 0.1, 3), (1500, 96, 0.04, 2)])
@pytest.mark.parametrize('unsafe', [False, True])
def test_forward_backward_bit_equal(nf, is_, scale, ts, unsafe):
    rng = np.random.default_rng(nf + is_)
    faces = random_soup(rng, 1, nf, scale)
    tex = rng.uniform(0, 1, (1, nf, ts, ts, ts, 3)).astype(np.float32)
    a = rn.forward(faces, tex, is_, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, True, unsafe=unsafe, impl='oracle')
    b = rn.forward(faces, tex, is_, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, True, unsafe=unsafe, impl='ref')
    assert (a.face_index_map >= 0).sum() > 0
    for k in MAPS:
        assert biteq(getattr(a, k), getattr(b, k)), k
    g_rgb = rng.normal(size=a.rgb_map.shape).astype(np.float32)
    g_a = rng.normal(size=a.alpha_map.shape).astype(np.float32)
    g_d = rng.normal(size=a.depth_map.shape).astype(np.float32)
    ga, gb = rn.backward(a, g_rgb, g_a, g_d), rn.backward(b, g_rgb, g_a, g_d)
    assert biteq(ga[0], gb[0])
    assert biteq(ga[1], gb[1])


def test_multi_batch_without_textures():
    # bs > 1 is only well defined without rgb (rasterize.py:390 drops the batch offset in K4)
    rng = np.random.default_rng(7)
    faces = random_soup(rng, 3, 200, 0.15)
    a = rn.forward(faces, None, 40, 0.1, 100, 1e-4, None, False, True, True, impl='oracle')
    b = rn.forward(faces, None, 40, 0.1, 100, 1e-4, None, False, True, True, impl='ref')
    for k in ('face_index_map', 'weight_map', 'depth_map', 'face_inv_map', 'alpha_map'):
        assert biteq(getattr(a, k), getattr(b, k)), k
    g = rng.normal(size=a.alpha_map.shape).astype(np.float32)
    assert biteq(rn.backward(a, None, g, g)[0], rn.backward(b, None, g, g)[0])


def test_near_far_literals():
    # the reference pastes near/far into the kernel text: comparisons happen in double
    rng = np.random.default_rng(11)
    faces = random_soup(rng, 1, 120, 0.3, zlo=0.05, zhi=3.0)
    for near, far in ((0.1, 100), (0.5, 2.0), (0.3, 1.7)):
        a = rn.forward(faces, None, 32, near, far, 1e-4, None, False, True, True, impl='oracle')
        b = rn.forward(faces, None, 32, near, far, 1e-4, None, False, True, True, impl='ref')
        assert biteq(a.depth_map, b.depth_map) and biteq(a.face_index_map, b.face_index_map)
