#!/usr/bin/env python3
"""How far is the SAFE rasterizer (the product's contract: rasterize.py:238-360) from the reference's DEFAULT, the
"unsafe" scanline kernel (scripts/env.sh:11 sets NEURAL_RENDERER_UNSAFE=1 -> rasterize.py:102-236)?

Both paths exist in the CPU oracle (oracle/raster_oracle.c restates K1 and K2+K3; bit-equal to the reference's own kernel
strings compiled for the CPU, tests/test_oracle_vs_ref.py).  This script renders the configs[1] object -- a car-like mesh
posed by PerspectiveTransform, silhouette / normal / depth maps at R with 2x anti-aliasing -- through
derender3d's Renderer semantics (oracle/nr_oracle.SDNRenderer) with each path and reports, at the R x R map level a
reference user sees: pixels whose value differs, maximum absolute differences, and the same at the S x S face-index
level.  K1's tie winner depends on the GPU's scheduling in the reference; the oracle executes faces in index order.

    python tools/safe_vs_unsafe.py [--tris 45000] [--render-size 384] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def compare(n_tris=45000, render_size=384, seed=1):
    from oracle import nr_oracle as no
    from oracle import raster_np as rn
    from sdn_hip import synth
    from util import posed_mesh
    v, f = synth.car_like(n_tris, seed=seed)
    pv, ang = posed_mesh(v, f, render_size=render_size)
    vt, ft = torch.tensor(pv), torch.tensor(f[None])
    maps = {}
    secs = {}
    for name, kw in (('safe', {}), ('unsafe', {'unsafe': True})):
        r = no.SDNRenderer(image_size=render_size, viewing_angle=ang)
        r.raster_kw = kw
        t0 = time.time()
        with torch.no_grad():
            maps[name] = (r(vt, ft, render_type=no.RenderType.Silhouette)[0, 0].numpy(),
                          r(vt, ft, render_type=no.RenderType.Normal)[0].numpy(),
                          r(vt, ft, render_type=no.RenderType.Depth)[0, 0].numpy())
        secs[name] = time.time() - t0
    (ma, na, da), (mb, nb, db) = maps['safe'], maps['unsafe']
    R = render_size
    covered = int((ma > 0).sum())
    out = {
        'triangles': int(len(f)), 'faces_with_fill_back': int(2 * len(f)), 'render_size': R, 'internal_size': 2 * R,
        'pixels': R * R, 'covered_pixels_safe': covered,
        'silhouette_pixels_differing': int((ma != mb).sum()), 'silhouette_max_abs': float(np.abs(ma - mb).max()),
        'normal_pixels_differing': int((np.abs(na - nb).max(0) > 1e-4).sum()), 'normal_max_abs': float(np.abs(na - nb).max()),
        'depth_pixels_differing_1e-4': int((np.abs(da - db) > 1e-4).sum()), 'depth_max_abs': float(np.abs(da - db).max()),
        'depth_max_abs_where_silhouettes_agree': float(np.abs(da - db)[ma == mb].max()),
        'oracle_threads': rn.num_threads(), 'seconds': secs,
    }
    out['silhouette_fraction_of_covered'] = out['silhouette_pixels_differing'] / max(covered, 1)
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--tris', type=int, default=45000)
    ap.add_argument('--render-size', type=int, default=384)
    ap.add_argument('--json')
    a = ap.parse_args()
    res = compare(a.tris, a.render_size)
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, 'w') as fh:
            json.dump(res, fh, indent=1)
