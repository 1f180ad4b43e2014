#!/usr/bin/env python3
"""Generate tests/golden/cad_golden_hi.npz: the OTHER five ShapeNet CAD meshes the reference ships, at the resolution the
reference renders them (scripts/main.py:44 `render_size` 384, 768^2 internal), through the REFERENCE's own kernel strings.

tests/golden/cad_golden.npz (make_cad_golden.py) holds all six meshes at R 192 and config 2's mesh a0fe4aac... at R 384
(`hi/`).  VERDICT r05 missing #3: m1..m5 at R 384 were untested, and mesh 2 (3776e4d1..., 45 056 triangles) is the slowest
mesh of the set by 2 x -- overflow lists, wave-shared boxes and the thin-face band path see different populations at 768^2.
For each of m1..m5 this script repeats make_cad_golden.py's steps 1-4 at R = 384: reference `load_obj`,
`ShapenetObj.__init__`'s normalisation, config 2's pose, SAFE path (K2 + K3 of rasterize.py:238-360) silhouette / normal /
depth through oracle/nr_oracle.py with impl='ref' (oracle/_ref/libnr_ref.so = the reference's own kernel strings), and
the gradient of the silhouette loss of scripts/main.py:445-451 against a fixed box target (K5, rasterize.py:523-745).
The K1 maps are not repeated here (K1 at R 384 is pinned on config 2's mesh, `hi/k1_*`).

The brute-force kernels are serial: ~15-25 minutes per mesh on one core, so the meshes run as separate processes:

    for k in 1 2 3 4 5; do python tests/golden/make_cad_golden_hi.py --mesh $k & done; wait
    python tests/golden/make_cad_golden_hi.py --merge --templates

Stored per mesh `m<k>/`: posed vertices (faces are cad_golden.npz's `m<k>/faces`), viewing angle, the S x S face-index map
(int32), the three R x R maps, the vertex gradient.  tests/test_cad_golden.py (CPU) and tests/test_gpu_cad_golden.py (HIP path against
it) read it.  `--templates` writes tests/golden/cad_templates.npz: all six templates as `ShapenetObj` keeps them (vertices
before any pose + faces), which bench.py decodes and poses for `value_real_meshes`.  Runs only where /root/reference exists.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_cad_golden as base  # noqa: E402  (sets sys.path for oracle / tests / product packages)

R_HI = base.R_HI
BOX = base.TARGET_BOX_HI
PART = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'cad_golden_hi_m%d.npz')
OUT = os.path.join(HERE, 'cad_golden_hi.npz')
NAMES = ('mask', 'normal', 'depth', 'face_index', 'grad')


def one(k):
    from oracle import raster_np as rn
    from util import posed_mesh
    assert rn.have_ref(), 'oracle/_ref/libnr_ref.so missing: run oracle/build_ref.py where /root/reference exists'
    cls, oid = base.MESHES[k]
    t0 = time.time()
    v, f = base.reference_load_obj()(os.path.join(base.ASSETS, cls, oid, 'models', 'model_normalized.obj'))
    v = base.shapenet_obj(v).astype(np.float32)
    pv, ang = posed_mesh(v, f, render_size=R_HI)
    safe = base.render(pv, f, ang, {'impl': 'ref'}, True, R_HI, BOX)
    out = {'verts': pv[0], 'angle': np.float64(ang), 'nfaces': np.int32(len(f))}
    for name in NAMES:
        out[name] = safe[name]
    np.savez_compressed(PART % k, **out)
    print('m%d %s/%s at R %d: %d triangles, %d covered pixels (%.0f s)' % (k, cls, oid, R_HI, len(f),
                                                                        int((safe['mask'] > 0).sum()), time.time() - t0), flush=True)


def merge():
    out = {'render_size': np.int32(R_HI), 'target_box': np.asarray(BOX, np.int32),
           'meshes': np.asarray(['%s/%s' % m for m in base.MESHES])}
    for k in range(1, 6):
        d = np.load(PART % k)
        for name in d.files:
            out['m%d/%s' % (k, name)] = d[name]
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


def templates():
    """tests/golden/cad_templates.npz: the six TEMPLATES as Derenderer3d holds them (`ShapenetObj.__init__`,
    derender3d/models/__init__.py:29-31, on the reference's `load_obj`) -- vertices before any pose + int32 faces.  What bench.py
    decodes and poses for `value_real_meshes` (nothing on the GPU box has the OBJ files)."""
    out = {'meshes': np.asarray(['%s/%s' % m for m in base.MESHES])}
    load_ref = base.reference_load_obj()
    for k, (cls, oid) in enumerate(base.MESHES):
        v, f = load_ref(os.path.join(base.ASSETS, cls, oid, 'models', 'model_normalized.obj'))
        out['t%d/verts' % k] = base.shapenet_obj(v).astype(np.float32)
        out['t%d/faces' % k] = f.astype(np.int32)
    path = os.path.join(HERE, 'cad_templates.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--mesh', type=int, choices=range(1, 6))
    ap.add_argument('--merge', action='store_true')
    ap.add_argument('--templates', action='store_true')
    a = ap.parse_args()
    if a.mesh:
        one(a.mesh)
    if a.merge:
        merge()
    if a.templates:
        templates()
