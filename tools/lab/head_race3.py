"""Lab: sdn_conv_wgrad_narrow on a SIDE stream while the main stream runs sdn_conv_head_mfma (the head data-gradient shape, 16 -> 64
channels in four row groups) -- on the SAME dz, or on a private copy of it."""
import ctypes
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd')]
import sdn_hip  # noqa: E402
if os.environ.get('LAB_LIB'):
    sdn_hip.LIB_PATH = os.path.abspath(os.environ['LAB_LIB'])
    print('library', sdn_hip.LIB_PATH)
from sdn_hip import check, lib, ptr  # noqa: E402
from sdn_hip import conv as hc  # noqa: E402
from sdn_hip import convplan as cp  # noqa: E402

N, H, W, C = 4, 192, 624, 64
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device='cuda')
dz = torch.zeros(N, H, W, 16, device='cuda')
dz[..., :4] = torch.randn(N, H, W, 4, device='cuda')
dz2 = dz.clone()
dy = (ctypes.c_int8 * 49)(*[k // 7 - 3 for k in range(49)])
dx = (ctypes.c_int8 * 49)(*[k % 7 - 3 for k in range(49)])
xp = F.pad(x.permute(0, 3, 1, 2).double().cpu(), (3, 3, 3, 3), mode='reflect')
R = 4
ref = torch.zeros(R, 49, C, dtype=torch.float64)
dzc = dz[..., :R].double().cpu()
for t in range(49):
    ky, kx = t // 7, t % 7
    ref[:, t] = torch.einsum('nhwr,nchw->rc', dzc, xp[:, :, ky:ky + H, kx:kx + W])
side = torch.cuda.Stream()
# head data-gradient weights: Conv2d(64, 4, 7): dgrad orientation rows = 64 input channels over dz's 16 padded channels
conv = nn.Conv2d(64, 4, 7, padding=3).cuda()
st = hc.Stage('conv', conv, 0, reflect=3)
launches, (GH, GW) = cp.conv_dgrad(7, 1, 3, H, W, True)
L = launches[0]
e = st.head_mfma('dgrad', L.taps, L.tapidx, 16, None)
e.refresh()
KH, KW, dy_min, dx_min, RR = e.meta
target = torch.empty(N, GH, GW, 64, device='cuda')
other_w = torch.randn(64 * 1024 * 1024, device='cuda')


def once(load):
    dw = torch.zeros(16, 49 * C, device='cuda')
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        check(lib().sdn_conv_wgrad_narrow(ptr(dz), ptr(x), ptr(dw), N, H, W, 16, R, H, W, C, 49, dy, dx, 1, 0, 1,
                                          ctypes.c_void_p(side.cuda_stream)))
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if load in ('dgrad_same_dz', 'dgrad_other_dz'):
        src = dz if load == 'dgrad_same_dz' else dz2
        for _ in range(2):
            check(lib().sdn_conv_head_mfma(ptr(src), N, H, W, 16, ptr(target), GH, GW, 64, RR, ptr(e.buf), KH, KW, dy_min, dx_min,
                                           0, 0, None, 0, None, main))
    elif load == 'stores':
        for _ in range(10):
            other_w.fill_(1.0)
    torch.cuda.synchronize()
    got = dw[:R].double().cpu().reshape(R, 49, C)
    xr = F.relu(xp)
    errs = [float((got[r] - ref_relu[r]).norm() / ref_relu[r].norm()) for r in range(R)]
    print('main stream load %-15s rows rel %s' % (load, ['%.1e' % v for v in errs]), flush=True)


xr = F.relu(xp)
ref_relu = torch.zeros(R, 49, C, dtype=torch.float64)
for t in range(49):
    ky, kx = t // 7, t % 7
    ref_relu[:, t] = torch.einsum('nhwr,nchw->rc', dzc, xr[:, :, ky:ky + H, kx:kx + W])
for load in ('none', 'dgrad_same_dz', 'dgrad_other_dz', 'stores', 'dgrad_same_dz', 'dgrad_other_dz', 'none'):
    once(load)
