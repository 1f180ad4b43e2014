"""Lab: kernels in which hipcc itself emits packed fp32 math with a HIGH-half splat (`v_pk_* ... op_sel:[1,...]`), beside an MFMA
kernel on another stream -- is the gfx950 packed-FMA finding of conv_narrow.hip (inline asm, src1 high splat) also true of the
compiler's own forms?
  (A) sdn_conv_narrow_fwd, 8 rows, 7 x 7, 64 channels (196 such instructions) against float64, alone / beside sdn_conv_head_mfma;
  (B) the silhouette backward of a 16-object frame (k_edge_rows): the vertex gradient alone vs beside the same neighbour.
usage: [LAB_LIB=lab/x.so] python tools/lab/pk_race.py"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')]
import sdn_hip  # noqa: E402
if os.environ.get('LAB_LIB'):
    sdn_hip.LIB_PATH = os.path.abspath(os.environ['LAB_LIB'])
print('library', sdn_hip.LIB_PATH)
from sdn_hip import check, lib, ptr  # noqa: E402
from sdn_hip import conv as hc  # noqa: E402
from sdn_hip import convplan as cp  # noqa: E402

DEV = 'cuda:0'
N, H, W = 4, 192, 624
torch.manual_seed(0)
# ---- the neighbour: head data gradient 16 -> 64 (four row groups), ~0.2 ms per launch
convn = nn.Conv2d(64, 4, 7, padding=3).to(DEV)
stn = hc.Stage('conv', convn, 0, reflect=3)
launches, (GH, GW) = cp.conv_dgrad(7, 1, 3, H, W, True)
en = stn.head_mfma('dgrad', launches[0].taps, launches[0].tapidx, 16, None)
en.refresh()
nKH, nKW, ndy, ndx, nRR = en.meta
dzn = torch.randn(N, H, W, 16, device=DEV)
target = torch.empty(N, GH, GW, 64, device=DEV)


def neighbour(reps):
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(reps):
        check(lib().sdn_conv_head_mfma(ptr(dzn), N, H, W, 16, ptr(target), GH, GW, 64, nRR, ptr(en.buf), nKH, nKW, ndy, ndx,
                                       0, 0, None, 0, None, main))


side = torch.cuda.Stream()
sidep = ctypes.c_void_p(side.cuda_stream)

# ---- (A)
cin, cout = 64, 8
conv = nn.Conv2d(cin, cout, 7, padding=0).to(DEV)
x = torch.randn(N, cin, H, W, device=DEV)
ref = F.conv2d(F.pad(x.double(), (3, 3, 3, 3), mode='reflect'), conv.weight.detach().double(), conv.bias.detach().double())
st = hc.Stage('conv', conv, 0, reflect=3)
fl, (OH, OW) = cp.conv_fwd(7, 1, 3, H, W)
nar = st.narrow('fwd', fl[0].taps, fl[0].tapidx, cin)
nar.refresh()
xg = x.permute(0, 2, 3, 1).contiguous()
bias = torch.zeros(16, device=DEV)
bias[:cout] = conv.bias.detach()
for load in (0, 3, 3, 3, 3, 0):
    out = torch.zeros(N, OH, OW, 16, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        check(lib().sdn_conv_narrow_fwd(ptr(xg), N, H, W, cin, ptr(out), OH, OW, 16, cout, ptr(nar.buf), 7, 7, -3, -3, 1, 0, ptr(bias),
                                        0, sidep))
    neighbour(load)
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).double()
    errs = [float((got[:, r] - ref[:, r]).abs().max() / ref[:, r].abs().max()) for r in range(cout)]
    print('(A) narrow_fwd 64 -> 8, neighbour launches %d: per output channel %s' % (load, ' '.join('%.1e' % e for e in errs)), flush=True)

# ---- (B)
import bench  # noqa: E402
device = torch.device('cuda', 0)
bank, sizes, cls, params, targets, ptf = bench.build_scene(device, seed=1234, mesh='cad_like')
step = bench.make_step(device, bank, cls, params, targets, ptf, backward=True, pack=False)


def grads(load):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        step()
    neighbour(load)
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in params.items()}


base = grads(0)
for load in (0, 6, 6, 6, 6, 0):
    g = grads(load)
    print('(B) frame step, neighbour launches %d: %s' % (load, '  '.join(
        '%s %.1e' % (k, float((g[k] - base[k]).abs().max() / base[k].abs().max())) for k in g)), flush=True)
