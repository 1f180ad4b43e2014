"""Lab: sdn_conv_wgrad_narrow at the benchmark's head shapes (bs 4, 384 x 1248), torch events around 10 launches.
Run once per setting of SDN_WGRAD_NARROW_ROW (read once per process): unset = row kernel for dense 7 x 7, 0 = column kernel."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, '3d-sdn_amd'))
from sdn_hip import check, lib, ptr, stream  # noqa: E402

DEV = 'cuda:0'
taps = [(dy, dx) for dy in range(-3, 4) for dx in range(-3, 4)]
dy = (ctypes.c_int8 * 49)(*[t[0] for t in taps])
dx = (ctypes.c_int8 * 49)(*[t[1] for t in taps])
for name, N, H, W, cin, R, relu in (('generator head 64 -> 3', 4, 384, 1248, 64, 3, 1), ('encoder head 16 -> 5', 4, 384, 1248, 16, 5, 0),
                                    ('inference-sized 64 -> 3', 4, 368, 1248, 64, 3, 1)):
    torch.manual_seed(1)
    x = torch.randn(N, H, W, cin, device=DEV)
    g = torch.randn(N, H, W, 16, device=DEV)
    dw = torch.zeros(16, 49 * cin, device=DEV)

    def go():
        check(lib().sdn_conv_wgrad_narrow(ptr(g), ptr(x), ptr(dw), N, H, W, 16, R, H, W, cin, 49, dy, dx, 1, 0, relu, stream()))
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    flop = 2.0 * N * H * W * 49 * cin * R
    print('%-28s SDN_WGRAD_NARROW_ROW=%s  %8.1f us  %6.1f TFLOP/s (real rows)  checksum %.6e' % (
        name, os.environ.get('SDN_WGRAD_NARROW_ROW', 'unset'), us, flop / us / 1e6, float(dw[:R].double().abs().sum()) / 13))
