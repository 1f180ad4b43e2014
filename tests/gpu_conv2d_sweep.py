"""Development aid (GPU): hb.conv2d forward / data gradient / weight gradient against F.conv2d in fp64 over a sweep of
ResNet shapes; prints the relative errors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd')):
    sys.path.insert(0, p)
import torch
import torch.nn as nn
import torch.nn.functional as F
from sdn_hip import bnnet as hb


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


cases = []
for N in (1, 3, 4, 8):
    for (cin, cout, k, s, p) in ((64, 64, 3, 1, 1), (64, 128, 3, 2, 1), (128, 128, 3, 1, 1), (64, 128, 1, 2, 0), (128, 256, 3, 2, 1),
                                 (256, 256, 3, 1, 1), (256, 512, 3, 2, 1), (512, 512, 3, 1, 1), (3, 64, 7, 2, 3)):
        for (H, W) in ((12, 14), (6, 7), (8, 8), (4, 6), (16, 16)):
            cases.append((N, cin, cout, k, s, p, H, W))
bad = 0
for (N, cin, cout, k, s, p, H, W) in cases:
    torch.manual_seed(1)
    m = nn.Conv2d(cin, cout, k, s, p, bias=False)
    x = torch.randn(N, cin, H, W)
    xr = x.double().requires_grad_(True)
    wr = m.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    m = m.cuda()
    xg = x.cuda().requires_grad_(True)
    y = hb.conv2d(m, xg)
    y.backward(gy.float().cuda())
    e = (rel(y, yr), rel(xg.grad, xr.grad), rel(m.weight.grad, wr.grad))
    flag = '' if max(e) < 5e-5 else '   <-----'
    bad += bool(flag)
    if flag or (N, H) == (4, 12):
        print('N %d %3d->%3d k%d s%d %2dx%2d  y %.1e dx %.1e dw %.1e%s' % (N, cin, cout, k, s, H, W, e[0], e[1], e[2], flag))
print('%d of %d cases off' % (bad, len(cases)))
