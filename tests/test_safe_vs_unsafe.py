"""CPU: what a reference user running the DEFAULT rasterizer (scripts/env.sh:11 -> NEURAL_RENDERER_UNSAFE=1 -> K1,
rasterize.py:102-236) gets, against the safe path the product implements (K2+K3, :238-360), at the map level the
callers see -- quantified, because the product's DEFAULT is the safe rule (K1 is the opt-in `use_unsafe_rasterizer` /
NEURAL_RENDERER_UNSAFE path since r04: tests/test_gpu_k1_coverage.py).  Both paths are the oracle's (bit-equal to the reference's kernel strings,
tests/test_oracle_vs_ref.py).  The full-size numbers (45k triangles, R = 384) are in profiles/r03_safe_vs_unsafe.json,
produced by tools/safe_vs_unsafe.py; this test runs the same comparison at a size that takes seconds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'tools'), os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def test_closed_mesh_maps_agree_within_the_parity_gate():
    """A closed, consistently wound mesh with fill_back: interior edges are shared, so the two coverage rules (pixel
    centre ON an edge belongs to both neighbours / to the scanline owner) can only disagree on the outline, and the 2x2
    average pooling halves what is left: the R x R maps must agree within the 1e-4 gate except on a sliver of outline pixels."""
    import safe_vs_unsafe as svu
    r = svu.compare(4000, 96)
    assert r['covered_pixels_safe'] > 2000
    assert r['silhouette_pixels_differing'] <= 0.005 * r['covered_pixels_safe'], r
    assert r['normal_pixels_differing'] <= 0.005 * r['covered_pixels_safe'], r
    assert r['depth_max_abs_where_silhouettes_agree'] <= 1e-4, r


def test_triangle_soup_differs_only_on_edge_pixels():
    """Open triangles (no neighbour to cover an edge): the two rules differ exactly where a pixel centre meets an edge or
    a tie is resolved by order; everywhere else face index and depth agree."""
    from oracle import raster_np as rn
    from util import random_soup
    rng = np.random.default_rng(11)
    faces = random_soup(rng, 1, 300, 0.15)
    a = rn.forward(faces, None, 96, 0.1, 100, 1e-4, None, False, True, True)
    b = rn.forward(faces, None, 96, 0.1, 100, 1e-4, None, False, True, True, unsafe=True)
    same = a.face_index_map == b.face_index_map
    assert 0.97 < same.mean() <= 1.0
    np.testing.assert_allclose(a.depth_map[same], b.depth_map[same], rtol=5e-4)
    # the disagreeing pixels sit on triangle outlines: in one of the two maps an 8-neighbour carries another face index
    ys, xs = np.nonzero(~same[0])
    fa, fb = a.face_index_map[0], b.face_index_map[0]
    assert 0 < len(ys) < 40
    for y, x in zip(ys, xs):
        y0, y1, x0, x1 = max(y - 1, 0), min(y + 2, 96), max(x - 1, 0), min(x + 2, 96)
        assert (fa[y0:y1, x0:x1] != fa[y, x]).any() or (fb[y0:y1, x0:x1] != fb[y, x]).any(), (y, x)
