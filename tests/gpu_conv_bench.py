#!/usr/bin/env python3
"""Micro-benchmark of the textural conv stack on the GPU box (development aid, not a test):
per-network forward / forward+backward times at the BASELINE size and a per-layer breakdown of the generator."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def timeit(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from models import networks as N
    bs = int(os.environ.get('BS', '4'))
    H, W = int(os.environ.get('H', '384')), int(os.environ.get('W', '1248'))
    torch.manual_seed(0)
    G = N.define_G(48, 3, 64, 'global', 4, 9).cuda()
    D = N.define_D(18, 64, 3, 'instance', False, 3, True).cuda()
    x = torch.randn(bs, 48, H, W, device='cuda')
    xd = torch.randn(bs, 18, H, W, device='cuda')
    gflop_g = 930.6 * bs * (H * W) / (384 * 1248)
    gflop_d = 71.5 * bs * (H * W) / (384 * 1248)

    def g_fwd():
        with torch.no_grad():
            return G(x)

    def g_fb():
        xx = x.clone().requires_grad_(True)
        G(xx).sum().backward()

    def d_fwd():
        with torch.no_grad():
            return D(xd)

    def d_fb():
        xx = xd.clone().requires_grad_(True)
        res = D(xx)
        sum(f.mean() for s in res for f in s).backward()

    for name, fn, gf in (('G fwd', g_fwd, gflop_g), ('G fwd+bwd', g_fb, 3 * gflop_g), ('D fwd', d_fwd, gflop_d),
                         ('D fwd+bwd', d_fb, 3 * gflop_d)):
        ms = timeit(fn)
        print('%-10s %8.2f ms   %7.1f TFLOP/s (algorithmic)' % (name, ms, gf / ms))
    sys.stdout.flush()


if __name__ == '__main__':
    main()
