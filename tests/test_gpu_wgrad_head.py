"""sdn_conv_wgrad_head_mfma (csrc/conv_whead.hip, r06): weight gradients of the 7 x 7 layers with <= 16 channels on the d(out) side
-- the generator head ReflectionPad2d(3) + Conv2d(64, 3, 7) (textural/models/networks.py:236), the encoder head 16 -> 5 (:306) and
the encoder stem 3 -> 16 (:291) -- on the matrix cores with bf16 x 3 split products.  Through the C ABI against the float64 sum

    dW[r, t, c] = sum_{n, y, x}  g[n, r, y, x] * pad(f(in))[n, c, y + dy_t, x + dx_t]

on ragged grids (tiles are 8 x 32 positions), with zero and reflected borders, ReLU on either operand, shuffled tap lists and every
rows_used from 1 to 16; refusals of what the kernel is not built for; and beside an MFMA kernel on another stream (the schedule
of the product: weight gradients on a side stream beside the data-gradient chain)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
_i8 = ctypes.c_int8
GATE = 2e-5   # bf16 x 3: the dropped lo * lo term is 2^-16 of a product; measured 2e-6 ... 6e-6

CASES = [  # name, N, H, W, cin (padded), rows_used, reflect, relu_rows, relu_gath, shuffled taps
    ('generator head 64 -> 3, reflect, ReLU on the input', 2, 24, 64, 64, 3, True, False, True, False),
    ('encoder head 16 -> 5, reflect, ragged grid', 1, 21, 45, 16, 5, True, False, False, False),
    ('encoder stem 3 (16) -> 16, reflect, ragged grid', 2, 19, 77, 16, 16, True, False, False, False),
    ('64 -> 4, zero border, shuffled tap list, ragged', 2, 13, 70, 64, 4, False, False, False, True),
    ('16 -> 8, zero border, ReLU on d(out)', 1, 16, 32, 16, 8, False, True, False, False),
    ('64 -> 1, reflect, one row of tiles narrower than a tile', 1, 7, 9, 64, 1, True, False, False, True),
    ('64 -> 16, zero border, many tiles per worker', 3, 200, 330, 64, 16, False, False, True, False),
    ('16 -> 11, reflect, many tiles per worker', 2, 260, 420, 16, 11, True, True, True, True),
]


def _taps(shuffled, seed):
    taps = [(dy, dx) for dy in range(-3, 4) for dx in range(-3, 4)]
    if shuffled:
        rng = np.random.default_rng(seed)
        taps = [taps[i] for i in rng.permutation(len(taps))]
    return taps


def _reference(x, g, taps, reflect, relu_rows, relu_gath):
    """float64 on the GPU (the big cases take seconds on the host): x [N, C, H, W], g [N, R, H, W] -> [R, taps, C]"""
    H, W = x.shape[2:]
    xr = (x.clamp(min=0) if relu_gath else x).double()
    gr = (g.clamp(min=0) if relu_rows else g).double()
    xp = F.pad(xr, (3, 3, 3, 3), mode='reflect') if reflect else F.pad(xr, (3, 3, 3, 3))
    return torch.stack([torch.einsum('nryx,ncyx->rc', gr, xp[:, :, 3 + dy:3 + dy + H, 3 + dx:3 + dx + W]) for dy, dx in taps], 1)


def _run(case):
    from sdn_hip import check, lib, ptr, stream
    name, N, H, W, cin, R, reflect, relu_rows, relu_gath, shuffled = CASES[case]
    taps = _taps(shuffled, 40 + case)
    torch.manual_seed(1900 + case)
    x = torch.randn(N, cin, H, W, device=DEV)
    g = torch.randn(N, R, H, W, device=DEV)
    ref = _reference(x, g, taps, reflect, relu_rows, relu_gath)
    xg = x.permute(0, 2, 3, 1).contiguous()
    gg = torch.full((N, H, W, 16), 7.0, device=DEV)       # channels behind rows_used must not reach rows < rows_used
    gg[..., :R] = g.permute(0, 2, 3, 1)
    dw = torch.zeros(16, len(taps) * cin, device=DEV)
    dy = (_i8 * len(taps))(*[t[0] for t in taps])
    dx = (_i8 * len(taps))(*[t[1] for t in taps])
    check(lib().sdn_conv_wgrad_head_mfma(ptr(gg), ptr(xg), ptr(dw), N, H, W, 16, R, H, W, cin, len(taps), dy, dx, int(reflect),
                                         int(relu_rows), int(relu_gath), stream()))
    torch.cuda.synchronize()
    assert float(dw[R:].abs().max()) == 0.0 if R < 16 else True, 'rows behind rows_used were written'
    got = dw[:R].reshape(R, len(taps), cin).double()
    return name, float((got - ref).abs().max()) / float(ref.abs().max())


@pytest.mark.parametrize('case', range(len(CASES)))
def test_head_weight_gradient_matches_float64(case):
    name, err = _run(case)
    assert err <= GATE, (name, err)


def test_accumulates_into_what_dw_holds():
    """the caller zeroes dw (the launch list's arena memset); the kernel only adds"""
    from sdn_hip import check, lib, ptr, stream
    N, H, W, cin, R = 1, 16, 40, 16, 5
    torch.manual_seed(5)
    x = torch.randn(N, H, W, cin, device=DEV)
    gg = torch.zeros(N, H, W, 16, device=DEV)
    gg[..., :R] = torch.randn(N, H, W, R, device=DEV)
    dy = (_i8 * 49)(*[k // 7 - 3 for k in range(49)])
    dx = (_i8 * 49)(*[k % 7 - 3 for k in range(49)])
    a = torch.zeros(16, 49 * cin, device=DEV)
    b = torch.full((16, 49 * cin), 3.0, device=DEV)
    for dw in (a, b):
        check(lib().sdn_conv_wgrad_head_mfma(ptr(gg), ptr(x), ptr(dw), N, H, W, 16, R, H, W, cin, 49, dy, dx, 1, 0, 0, stream()))
    torch.cuda.synchronize()
    assert float((b[:R] - 3.0 - a[:R]).abs().max()) <= 1e-4 * float(a.abs().max())
    assert float((b[R:] - 3.0).abs().max()) == 0.0


def test_refusals():
    from sdn_hip import lib, ptr, stream
    L = lib()
    x = torch.zeros(1, 16, 32, 64, device=DEV)
    g = torch.zeros(1, 16, 32, 16, device=DEV)
    dw = torch.zeros(16, 49 * 64, device=DEV)
    dy = (_i8 * 49)(*[k // 7 - 3 for k in range(49)])
    dx = (_i8 * 49)(*[k % 7 - 3 for k in range(49)])

    def call(Cr=16, R=3, Cc=64, ntaps=49, dyv=dy, dxv=dx, GH=16, GW=32, pad=1):
        return L.sdn_conv_wgrad_head_mfma(ptr(g), ptr(x), ptr(dw), 1, 16, 32, Cr, R, GH, GW, Cc, ntaps, dyv, dxv, pad, 0, 0, stream())
    assert call() == 0
    assert call(Cr=32) != 0 and b'16 channels' in L.sdn_last_error()
    assert call(Cc=32) != 0 and b'16 or 64' in L.sdn_last_error()
    assert call(R=0) != 0 and call(R=17) != 0
    assert call(ntaps=25) != 0 and b'49 taps' in L.sdn_last_error()
    hole = (_i8 * 49)(*([k // 7 - 3 for k in range(48)] + [-3]))       # the last tap repeats the first
    holx = (_i8 * 49)(*([k % 7 - 3 for k in range(48)] + [-3]))
    assert call(dyv=hole, dxv=holx) != 0 and b'dense' in L.sdn_last_error()
    assert call(GH=5, GW=32) != 0 and b'reflected border' in L.sdn_last_error()
    torch.cuda.synchronize()


def test_beside_an_mfma_kernel_on_another_stream():
    """The generator-head shape at 192 x 624, batch 4: the head kernel's 64-row data-gradient launch runs on the main stream while
    the weight gradient runs on a side stream, four rounds; every round within the gate and the rounds agree to float-atomic
    order."""
    from sdn_hip import check, lib, ptr
    from sdn_hip import conv as hc
    N, H, W, C, R = 4, 192, 624, 64, 3
    torch.manual_seed(78)
    x = torch.randn(N, H, W, C, device=DEV)
    dz = torch.zeros(N, H, W, 16, device=DEV)
    dz[..., :R] = torch.randn(N, H, W, R, device=DEV)
    taps = _taps(False, 0)
    dy = (_i8 * 49)(*[t[0] for t in taps])
    dx = (_i8 * 49)(*[t[1] for t in taps])
    ref = _reference(x.permute(0, 3, 1, 2), dz[..., :R].permute(0, 3, 1, 2), taps, True, False, True)
    # the neighbour: the head kernel's data gradient 16 -> 64 of the same layer (its 64-row form), on a private copy of d(out)
    import torch.nn as nn
    from sdn_hip import convplan as cp
    conv = nn.Conv2d(64, R, 7, padding=3).to(DEV)
    st = hc.Stage('conv', conv, 0, reflect=3)
    launches, (GH, GW) = cp.conv_dgrad(7, 1, 3, H, W, True)
    e = st.head_mfma('dgrad', launches[0].taps, launches[0].tapidx, 16, None)
    e.refresh()
    KH, KW, dy_min, dx_min, RR = e.meta
    target = torch.empty(N, GH, GW, 64, device=DEV)
    dz2 = dz.clone()
    side = torch.cuda.Stream()
    errs = []
    for _ in range(4):
        dw = torch.zeros(16, 49 * C, device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            check(lib().sdn_conv_wgrad_head_mfma(ptr(dz), ptr(x), ptr(dw), N, H, W, 16, R, H, W, C, 49, dy, dx, 1, 0, 1,
                                                 ctypes.c_void_p(side.cuda_stream)))
        main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            check(lib().sdn_conv_head_mfma(ptr(dz2), N, H, W, 16, ptr(target), GH, GW, 64, RR, ptr(e.buf), KH, KW, dy_min, dx_min,
                                           0, 0, None, 0, None, main))
        torch.cuda.synchronize()
        got = dw[:R].reshape(R, 49, C).double()
        errs.append(float((got - ref).abs().max()) / float(ref.abs().max()))
    assert max(errs) <= GATE, errs
