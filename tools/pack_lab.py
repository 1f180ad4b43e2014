#!/usr/bin/env python3
"""Development aid (GPU): the weight pack / gradient unpack kernels alone, on the residual-layer weight of the generator
(1024 x 1024 x 3 x 3, networks.py:261) and a transposed-conv weight, through the C ABI -- microseconds and GB/s of the bytes each
call must move (fp32 source read + bf16 hi/lo written, or fp32 read + fp32 written)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    from sdn_hip import check, lib, ptr, stream
    from sdn_hip import convplan as cp
    L = lib()
    dev = 'cuda:0'
    for name, shape, transposed in (('conv 1024x1024x3x3', (1024, 1024, 3, 3), False), ('convT 1024x512x3x3', (1024, 512, 3, 3), True),
                                    ('conv 512x256x4x4', (512, 256, 4, 4), False)):
        w = torch.randn(*shape, device=dev)
        k = shape[2]
        kk = k * k
        if not transposed:
            cout, cin = shape[0], shape[1]
            orient = {'fwd': (cout, cin, cin * kk, kk), 'dgrad': (cin, cout, kk, cin * kk)}
        else:
            cin, cout = shape[0], shape[1]
            orient = {'fwd': (cout, cin, kk, cout * kk), 'dgrad': (cin, cout, cout * kk, kk)}
        tix = torch.arange(kk, dtype=torch.int32, device=dev)
        for which, (R, C, sr, sc) in orient.items():
            Ccp = cp.cpad(C)
            rows_g, rows_t = cp.weight_rows(cp.cpad(R)), cp.tile_weight_rows(cp.cpad(R))
            Kp = cp.kpad(kk, Ccp)
            pk = torch.empty(2 * rows_g * Kp, dtype=torch.bfloat16, device=dev)
            pt = torch.empty(2 * rows_t * kk * Ccp, dtype=torch.bfloat16, device=dev)
            dw = torch.randn(R, kk * Ccp, device=dev)
            gw = torch.empty_like(w)
            byt = w.numel() * 4 * 2
            t1 = timed(lambda: check(L.sdn_conv_pack_weights(ptr(w), R, C, sr, sc, ptr(tix), kk, Ccp, Kp, rows_g, ptr(pk), stream())))
            t2 = timed(lambda: check(L.sdn_conv_pack_weights_kmajor(ptr(w), R, C, sr, sc, ptr(tix), kk, Ccp, rows_t, ptr(pt), stream())))
            t3 = timed(lambda: check(L.sdn_conv_unpack_grad(ptr(dw), R, C, sr, sc, ptr(tix), kk, Ccp, ptr(gw), 0, stream())))
            print('%-20s %-5s strides (row %d, col %d): pack %.1f us (%.0f GB/s)  pack k-major %.1f us (%.0f GB/s)  unpack %.1f us (%.0f GB/s)'
                  % (name, which, sr, sc, t1, byt / t1 / 1e3, t2, byt / t2 / 1e3, t3, byt / t3 / 1e3))
        t0 = timed(lambda: gw.copy_(w))
        print('%-20s copy_ of the same tensor: %.1f us (%.0f GB/s)' % (name, t0, w.numel() * 8 / t0 / 1e3))


if __name__ == '__main__':
    main()
