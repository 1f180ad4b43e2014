"""Lab: sdn_conv_wgrad_head_mfma beside the kernels it replaces, alone on the chip, at the product's shapes.
usage: python tools/lab/whead_time.py [N H W]      (default 4 192 624: one GPU's share of the textural bench batch)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd')]
from sdn_hip import check, lib, ptr, stream  # noqa: E402

N, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (4, 192, 624)
DEV = 'cuda:0'
dy = (ctypes.c_int8 * 49)(*[k // 7 - 3 for k in range(49)])
dx = (ctypes.c_int8 * 49)(*[k % 7 - 3 for k in range(49)])


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, C, R in (('generator head 64 -> 3', 64, 3), ('encoder head 16 -> 5', 16, 5), ('encoder stem 3 -> 16', 16, 16)):
    x = torch.randn(N, H, W, C, device=DEV)
    g = torch.zeros(N, H, W, 16, device=DEV)
    g[..., :R] = torch.randn(N, H, W, R, device=DEV)
    dw = torch.zeros(16, 49 * C, device=DEV)
    L = lib()
    flop = 2.0 * N * H * W * 49 * R * C
    t_new = timed(lambda: check(L.sdn_conv_wgrad_head_mfma(ptr(g), ptr(x), ptr(dw), N, H, W, 16, R, H, W, C, 49, dy, dx, 1, 0, 1, stream())))
    line = '%-24s N %d %dx%d  head mfma %.3f ms (%.1f TFLOP/s useful, %.1f issued bf16)' % (
        name, N, H, W, t_new, flop / t_new / 1e9, 3 * 2.0 * N * H * W * 49 * 16 * C / t_new / 1e9)
    if R <= 8:
        t_old = timed(lambda: check(L.sdn_conv_wgrad_narrow(ptr(g), ptr(x), ptr(dw), N, H, W, 16, R, H, W, C, 49, dy, dx, 1, 0, 1, stream())))
        line += '   narrow (fp32 VALU) %.3f ms' % t_old
    print(line, flush=True)
