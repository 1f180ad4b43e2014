"""Analytic known-answer tests of the oracle (the reference ships none for this path)."""
import numpy as np
import torch

from oracle import nr_oracle as no
from oracle import raster_np as rn


def test_single_triangle_depth_is_perspective_correct():
    # one big triangle with different vertex depths: zp = 1 / sum(w_k / z_k) with pixel-space barycentrics
    faces = np.array([[[[-0.8, -0.8, 1.0], [0.8, -0.8, 2.0], [0.0, 0.9, 4.0]]]], np.float32)
    is_ = 64
    st = rn.forward(faces, None, is_, 0.1, 100, 1e-4, None, False, True, True)
    cov = st.face_index_map[0] >= 0
    assert cov.sum() > 800
    ys, xs = np.nonzero(cov)
    p = 0.5 * (faces[0, 0, :, :2].astype(np.float64) * is_ + is_ - 1)
    M = np.array([[p[0, 0], p[1, 0], p[2, 0]], [p[0, 1], p[1, 1], p[2, 1]], [1, 1, 1]])
    w = np.linalg.solve(M, np.stack([xs, ys, np.ones_like(xs)]).astype(np.float64))
    zp = 1.0 / (w[0] / 1.0 + w[1] / 2.0 + w[2] / 4.0)
    np.testing.assert_allclose(st.depth_map[0][cov], zp, rtol=2e-5)
    np.testing.assert_allclose(st.weight_map[0][cov], w.T, atol=2e-5)
    assert np.all(st.alpha_map[0][cov] == 1) and np.all(st.alpha_map[0][~cov] == 0)
    assert np.all(st.depth_map[0][~cov] == 100)


def test_backface_is_culled_and_fill_back_restores_it():
    tri = np.array([[[[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.5, 1.0]]]], np.float32)  # counter-clockwise
    cw = tri[:, :, ::-1].copy()
    assert (rn.forward(tri, None, 32, 0.1, 100, 1e-4, None, False, True, False).alpha_map.sum()) > 0
    assert (rn.forward(cw, None, 32, 0.1, 100, 1e-4, None, False, True, False).alpha_map.sum()) == 0
    both = np.concatenate([cw, cw[:, :, ::-1]], 1)
    assert (rn.forward(both, None, 32, 0.1, 100, 1e-4, None, False, True, False).face_index_map.max()) == 1


def test_z_tie_goes_to_lowest_face_index():
    tri = np.array([[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.5, 1.0]], np.float32)
    faces = np.stack([tri, tri, tri])[None]
    st = rn.forward(faces, None, 32, 0.1, 100, 1e-4, None, False, True, False)
    assert set(np.unique(st.face_index_map)) == {-1, 0}


def test_near_far_reject():
    tri = lambda z: np.array([[-0.5, -0.5, z], [0.5, -0.5, z], [0.0, 0.5, z]], np.float32)
    faces = np.stack([tri(0.05), tri(150.0), tri(3.0)])[None]
    st = rn.forward(faces, None, 32, 0.1, 100, 1e-4, None, False, True, True)
    assert set(np.unique(st.face_index_map)) == {-1, 2}
    np.testing.assert_allclose(st.depth_map[st.face_index_map >= 0], 3.0, rtol=1e-6)


def test_axis_aligned_cube_depth_plateau_normals_and_area():
    from sdn_hip import synth  # numpy-only mesh helper
    v, f = synth.cube()
    verts = torch.tensor((v + np.array([0, 0, -3.0], np.float32))[None])  # cube centred 3 in front of the camera
    r = no.SDNRenderer(image_size=64, viewing_angle=30)
    faces = torch.tensor(f[None])
    depth = r(verts, faces, render_type=no.RenderType.Depth)[0, 0].numpy()
    mask = r(verts, faces, render_type=no.RenderType.Silhouette)[0, 0].numpy()
    normal = r(verts, faces, render_type=no.RenderType.Normal)[0].numpy()
    inside = mask == 1
    # front face at distance 2.5 (the look-basis carries chainer's 1e-5 normalisation epsilon)
    np.testing.assert_allclose(depth[inside], 2.5, rtol=1e-4)
    # projected half-width of the front face: 0.5 / 2.5 / tan(30 deg) of the half image
    half = 0.5 / 2.5 / np.tan(30 / 180. * 3.1416)
    np.testing.assert_allclose(mask.sum(), (half * 64) ** 2, rtol=0.02)
    # front face normal points to the camera: (0, 0, 1) in the renderer's output convention up to sign conventions
    n = normal[:, inside]
    assert np.allclose(np.abs(n[2]), 1, atol=1e-4) and np.allclose(n[:2], 0, atol=1e-4)


def test_depth_gradient_matches_finite_differences():
    rng = np.random.default_rng(3)
    base = np.array([[[[-0.7, -0.6, 1.5], [0.8, -0.7, 2.5], [0.1, 0.8, 3.5]]]], np.float32)
    g = rng.normal(size=(1, 48, 48)).astype(np.float32)
    st = rn.forward(base, None, 48, 0.1, 100, 1e-4, None, False, False, True)
    cov = st.face_index_map >= 0
    gz = g * cov  # only interior pixels carry an analytic depth gradient
    gf, _ = rn.backward(st, None, None, gz)
    for k in range(3):  # z of each vertex
        h = 1e-3
        fp, fm = base.copy(), base.copy()
        fp[0, 0, k, 2] += h
        fm[0, 0, k, 2] -= h
        dp = rn.forward(fp, None, 48, 0.1, 100, 1e-4, None, False, False, True).depth_map
        dm = rn.forward(fm, None, 48, 0.1, 100, 1e-4, None, False, False, True).depth_map
        fd = ((dp - dm) / (2 * h) * gz).sum()
        np.testing.assert_allclose(gf[0, 0, k, 2], fd, rtol=2e-2)


def test_safe_and_unsafe_agree_on_interior_pixels():
    from util import random_soup
    rng = np.random.default_rng(5)
    faces = random_soup(rng, 1, 200, 0.2)
    a = rn.forward(faces, None, 64, 0.1, 100, 1e-4, None, False, True, True)
    b = rn.forward(faces, None, 64, 0.1, 100, 1e-4, None, False, True, True, unsafe=True)
    same = a.face_index_map == b.face_index_map
    assert same.mean() > 0.97  # the two coverage rules differ only on edge pixels
    np.testing.assert_allclose(a.depth_map[same], b.depth_map[same], rtol=5e-4)  # different vertex order in face_inv


def test_pool_and_flip_layout():
    x = torch.arange(16.).reshape(1, 4, 4)
    p = no.pool2x2(x.flip(1))
    assert p.shape == (1, 2, 2)
    assert float(p[0, 0, 0]) == (8 + 9 + 12 + 13) / 4
