"""Kernel-level timing of the rasterizer on the benchmark mesh (run on the GPU box):  python tests/gpu_microbench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric'),
          os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import sdn_hip  # noqa: E402
from sdn_hip import ops, synth  # noqa: E402
from util import posed_mesh  # noqa: E402
from derender3d.models.renderer import Renderer  # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    sdn_hip.timing_enable(True)
    sdn_hip.timing_read()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms, k = sdn_hip.timing_read()
    sdn_hip.timing_enable(False)
    return e0.elapsed_time(e1) / n * 1e3, (ms / max(k, 1)) * 1e3


def main():
    degenerate = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    v, f = synth.car_like(45000, seed=100, degenerate=degenerate)
    pv, ang = posed_mesh(v, f)
    r = Renderer(image_size=384)
    r.viewing_angle = ang
    _r, vert = r._setup(torch.tensor(pv, device=dev))
    fi = torch.tensor(f[None], device=dev)
    faces9 = _r.gather(_r.project(vert), fi).contiguous()
    colors = _r.face_normal_colors(vert, fi).contiguous()
    print('faces', tuple(faces9.shape), 'degenerate', degenerate)
    for bs in (1, 4, 16):
        F = faces9.expand(bs, -1, -1, -1).contiguous()
        C = colors.expand(bs, -1, -1).contiguous()
        Fg = F.clone().requires_grad_(True)
        cases = {
            'alpha            ': lambda: ops.RasterizeMaps.apply(F, None, 384, True, 0.1, 100, 1e-4, None, False, True, False, None, False),
            'alpha+depth      ': lambda: ops.RasterizeMaps.apply(F, None, 384, True, 0.1, 100, 1e-4, None, False, True, True, None, False),
            'normal+alpha+dep ': lambda: ops.RasterizeMaps.apply(F, C, 384, True, 0.1, 100, 1e-3, (0, 0, 0), True, True, True, 1e-4, True),
            'same + SAVE_MAPS ': lambda: ops.RasterizeMaps.apply(Fg, C, 384, True, 0.1, 100, 1e-3, (0, 0, 0), True, True, True, 1e-4, True),
            'alpha, no AA 768 ': lambda: ops.RasterizeMaps.apply(F, None, 768, False, 0.1, 100, 1e-4, None, False, True, False, None, False),
        }
        for name, fn in cases.items():
            tot, tile = timed(fn)
            print('bs=%2d %s  call %8.1f us   k_raster_tiles %8.1f us  (%.1f us / object)' % (bs, name, tot, tile, tile / bs))
    # backward: silhouette loss only
    for bs in (1, 16):
        F = faces9.expand(bs, -1, -1, -1).contiguous().requires_grad_(True)
        C = colors.expand(bs, -1, -1).contiguous()
        target = torch.zeros(bs, 384, 384, device=dev)
        target[:, 120:270, 40:340] = 1

        def fb():
            F.grad = None
            rgb, a, d = ops.RasterizeMaps.apply(F, C, 384, True, 0.1, 100, 1e-3, (0, 0, 0), True, True, True, 1e-4, True)
            ((a - target) ** 2).mean().backward()
        tot, tile = timed(fb, n=10)
        print('bs=%2d fwd+bwd(silhouette loss)  call %8.1f us  (%.1f us / object)' % (bs, tot, tot / bs))


if __name__ == '__main__':
    main()
