#!/usr/bin/env python3
"""Development aid (GPU): k_raster_tiles launch time on a REAL ShapeNet CAD mesh of the reference (tests/golden/cad_golden.npz:
config 2's mesh posed at render_size 384 when the fixture holds `hi/`, else at its R 192), 16 copies per launch like a frame's
objects -- the face order of a real file, which the synthetic templates do not reproduce.  The hierarchical depth cull is
selected per PROCESS (SDN_RASTER_HIZ = 0 / 1 / 2): run once per setting and compare."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    import sdn_hip
    from derender3d.models.renderer import Renderer
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'cad_golden.npz'))
    out = {}
    for k in range(6):
        if k == 0 and 'hi/verts' in d.files:
            pv, ang, R = d['hi/verts'], float(d['hi/angle']), int(d['hi/render_size'])
        else:
            pv, ang, R = d['m%d/verts' % k], float(d['m%d/angle' % k]), int(d['render_size'])
        f = d['m%d/faces' % k]
        n = 16
        r = Renderer(image_size=R)
        r.viewing_angle = [ang] * n
        vt = torch.tensor(np.repeat(pv[None], n, 0), device='cuda:0')
        fi = torch.tensor(np.repeat(f[None], n, 0), device='cuda:0')
        with torch.no_grad():
            for _ in range(3):
                r.render_maps(vt, fi)
            torch.cuda.synchronize()
            sdn_hip.timing_enable(True)
            sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
            for _ in range(20):
                r.render_maps(vt, fi)
            torch.cuda.synchronize()
            ms, cnt, _ = sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
            sdn_hip.timing_enable(False)
        out['mesh %d (%d tris, R %d)' % (k, len(f), R)] = round(ms / max(cnt, 1) * 1e3, 1)
    print('SDN_RASTER_HIZ=%s k_raster_tiles us per 16-object launch: %s' % (os.environ.get('SDN_RASTER_HIZ', '(default)'), out))


if __name__ == '__main__':
    main()
