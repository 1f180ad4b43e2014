#!/usr/bin/env python3
"""Generate tests/golden/raster_golden.npz from the REFERENCE's rasterizer kernels.

The kernels are the CUDA strings of /root/reference/geometric/neural_renderer/rasterize.py, compiled for the
CPU by oracle/build_ref.py (oracle/_ref/libnr_ref.so) and driven through the same host sequence as
Rasterize.forward_gpu / backward_gpu (oracle/raster_np.py, impl='ref').  Runs only where /root/reference
exists; the fixture it writes travels with the repository.

Cases (all bs = 1 because rasterize.py:390 drops the batch offset when it samples textures):
  soup_small   60 random triangles, 32x32, rgb+alpha+depth, textures 2^3
  soup_mid     400 random triangles, 64x64, rgb+alpha+depth, textures 3^3
  slivers      200 thin / degenerate triangles (repeated vertices, collinear points), 48x48
  cube         the 12+12 fill_back faces of a posed cube, 64x64
Each case stores inputs, every forward map, and the gradients for fixed upstream gradients.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3d-sdn_amd')]

from oracle import raster_np as rn  # noqa: E402
from util import random_soup  # noqa: E402


def sliver_faces(rng, n):
    f = random_soup(rng, 1, n, 0.25)
    k = n // 4
    f[0, :k, 2] = f[0, :k, 1]                                   # repeated vertex: exactly degenerate
    t = rng.uniform(0, 1, (k, 1)).astype(np.float32)
    f[0, k:2 * k, 2, :2] = f[0, k:2 * k, 0, :2] * (1 - t) + f[0, k:2 * k, 1, :2] * t   # (nearly) collinear
    f[0, 2 * k:3 * k, 2, :2] = f[0, 2 * k:3 * k, 1, :2] + rng.normal(0, 1e-4, (k, 2)).astype(np.float32)
    return f


def cube_faces():
    from sdn_hip import synth
    from util import posed_mesh
    from oracle import nr_oracle as no
    import torch
    v, f = synth.cube()
    pv, ang = posed_mesh(v, f, theta=0.5, scale=(1, 1, 1), translation=(0.3, 0.2, -3.0), render_size=64)
    r = no.NRRenderer()
    r.viewing_angle = ang
    r.camera_mode = 'look'
    r.eye = torch.zeros(1, 3)
    r.camera_direction = torch.tensor([[0., 0., -1.]])
    r.up = torch.tensor([[0., 1., 0.]])
    vt = torch.tensor(pv) * torch.tensor([-1., 1., 1.])
    faces = no.vertices_to_faces(r._camera(vt), r._fill_back(torch.tensor(f[None])))
    return faces.numpy()


def main():
    if not rn.have_ref():
        rn.build()
    assert rn.have_ref(), 'oracle/_ref/libnr_ref.so missing: run oracle/build_ref.py where /root/reference exists'
    rng = np.random.default_rng(424242)
    cases = {
        'soup_small': (random_soup(rng, 1, 60, 0.3), 32, 2),
        'soup_mid': (random_soup(rng, 1, 400, 0.08), 64, 3),
        'slivers': (sliver_faces(rng, 200), 48, 2),
        'cube': (cube_faces(), 64, 2),
    }
    out = {}
    for name, (faces, is_, ts) in cases.items():
        nf = faces.shape[1]
        tex = rng.uniform(0, 1, (1, nf, ts, ts, ts, 3)).astype(np.float32)
        st = rn.forward(faces, tex, is_, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, True, impl='ref')
        g_rgb = rng.normal(size=st.rgb_map.shape).astype(np.float32)
        g_alpha = rng.normal(size=st.alpha_map.shape).astype(np.float32)
        g_depth = rng.normal(size=st.depth_map.shape).astype(np.float32)
        gf, gt = rn.backward(st, g_rgb, g_alpha, g_depth)
        # alpha-only backward (what rasterize_silhouettes differentiates, eps 1e-4)
        st_a = rn.forward(faces, None, is_, 0.1, 100, 1e-4, None, False, True, False, impl='ref')
        gf_a, _ = rn.backward(st_a, None, g_alpha, None)
        out[name + '/faces'] = faces
        out[name + '/textures'] = tex
        out[name + '/image_size'] = np.int32(is_)
        for k in ('face_index_map', 'weight_map', 'depth_map', 'face_inv_map', 'rgb_map', 'alpha_map',
                  'sampling_index_map', 'sampling_weight_map'):
            out[name + '/' + k] = getattr(st, k)
        out[name + '/g_rgb'], out[name + '/g_alpha'], out[name + '/g_depth'] = g_rgb, g_alpha, g_depth
        out[name + '/grad_faces'], out[name + '/grad_textures'] = gf, gt
        out[name + '/grad_faces_alpha_only'] = gf_a
        print(name, 'nf', nf, 'covered', int((st.face_index_map >= 0).sum()), 'nan grads', int(np.isnan(gf).sum()))
    np.savez_compressed(os.path.join(HERE, 'raster_golden.npz'), **out)
    print('wrote raster_golden.npz')


if __name__ == '__main__':
    main()
