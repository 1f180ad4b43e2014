"""Lab: sdn_conv_wgrad_narrow (generator-head shape) through the C ABI on a SIDE stream while the main stream runs (a) nothing,
(b) torch matmuls, (c) torch element-wise kernels over an unrelated buffer, (d) a reader of the same dz.  Against float64."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd')]
import sdn_hip  # noqa: E402
from sdn_hip import check, lib, ptr  # noqa: E402

N, H, W, C = 4, 192, 624, 64
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device='cuda')
dz = torch.zeros(N, H, W, 16, device='cuda')
dz[..., :3] = torch.randn(N, H, W, 3, device='cuda')
dy = (ctypes.c_int8 * 49)(*[k // 7 - 3 for k in range(49)])
dx = (ctypes.c_int8 * 49)(*[k % 7 - 3 for k in range(49)])
xp = F.pad(x.permute(0, 3, 1, 2).double().cpu(), (3, 3, 3, 3), mode='reflect')
ref = torch.zeros(3, 49, C, dtype=torch.float64)
dzc = dz[..., :3].double().cpu()
for t in range(49):
    ky, kx = t // 7, t % 7
    ref[:, t] = torch.einsum('nhwr,nchw->rc', dzc, xp[:, :, ky:ky + H, kx:kx + W])
side = torch.cuda.Stream()
big_a = torch.randn(4096, 4096, device='cuda')
other = torch.randn(64 * 1024 * 1024, device='cuda')


def once(load):
    dw = torch.zeros(16, 49 * C, device='cuda')
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        check(lib().sdn_conv_wgrad_narrow(ptr(dz), ptr(x), ptr(dw), N, H, W, 16, 3, H, W, C, 49, dy, dx, 1, 0, 0,
                                          ctypes.c_void_p(side.cuda_stream)))
    if load == 'matmul':
        for _ in range(6):
            big_a @ big_a
    elif load == 'elementwise':
        for _ in range(20):
            other.mul_(1.0001)
    elif load == 'reader':
        for _ in range(20):
            (dz * 2.0).sum()
    torch.cuda.synchronize()
    got = dw[:3].double().cpu().reshape(3, 49, C)
    e = [float((got[r] - ref[r]).norm() / ref[r].norm()) for r in range(3)]
    print('main stream load %-12s rows rel %s' % (load, ['%.1e' % v for v in e]), flush=True)


for load in ('none', 'matmul', 'elementwise', 'reader', 'none', 'matmul', 'elementwise'):
    once(load)
