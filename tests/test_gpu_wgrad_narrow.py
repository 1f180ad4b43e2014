"""sdn_conv_wgrad_narrow (csrc/conv_narrow.hip): weight gradients of the 1-8 output-channel head layers -- the generator's
ReflectionPad2d(3) + Conv2d(64, 3, 7) (textural/models/networks.py:236) and the encoder's 16 -> 5 (:306) -- as run by
loss_G.backward() (textural/train.py:88-95).  Exact fp32 on the vector ALUs.  r05 added a second kernel behind the same entry
point (`k_wgrad_narrow_row`: dense 7 x 7 windows, one LDS read per 7 R FMAs); both are compared here, through the C ABI, with
the float64 sum

    dW[r, t, c] = sum_{n, y, x}  g[n, r, y, x] * pad(f(in))[n, c, y + dy_t, x + dx_t]

on ragged grids, with zero and reflected borders, ReLU on either operand, shuffled tap lists (the kernel must honour the
caller's tap order) and a window with a hole (which the row kernel must leave to the column kernel)."""
import ctypes
import os
import subprocess
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

CASES = [  # name, N, H, W, cin, rows_used, reflect, relu_rows, relu_gath, taps ('dense' | 'shuffled' | 'hole' | '5x5')
    ('generator head 64 -> 3, reflect, ReLU on the input', 2, 24, 64, 64, 3, True, False, True, 'dense'),
    ('encoder head 16 -> 5, reflect, ragged grid', 1, 21, 45, 16, 5, True, False, False, 'dense'),
    ('64 -> 4, zero border, shuffled tap list, ragged', 2, 13, 70, 64, 4, False, False, False, 'shuffled'),
    ('16 -> 8, zero border, ReLU on d(out)', 1, 16, 32, 16, 8, False, True, False, 'dense'),
    ('32 -> 2, reflect, one row of tiles', 1, 5, 40, 32, 2, True, False, False, 'shuffled'),
    ('16 -> 3, a 7 x 7 window with a hole (column kernel)', 1, 12, 33, 16, 3, False, False, False, 'hole'),
    ('16 -> 3, 5 x 5 window (column kernel)', 1, 12, 33, 16, 3, True, False, False, '5x5'),
]


def _taps(kind, seed):
    k = 5 if kind == '5x5' else 7
    h = k // 2
    taps = [(dy, dx) for dy in range(-h, h + 1) for dx in range(-h, h + 1)]
    if kind == 'shuffled':
        rng = np.random.default_rng(seed)
        taps = [taps[i] for i in rng.permutation(len(taps))]
    if kind == 'hole':
        taps.remove((1, -2))
    return taps, h


def _run(case):
    from sdn_hip import check, lib, ptr, stream
    name, N, H, W, cin, R, reflect, relu_rows, relu_gath, kind = CASES[case]
    taps, h = _taps(kind, 40 + case)
    torch.manual_seed(900 + case)
    x = torch.randn(N, cin, H, W)
    g = torch.randn(N, R, H, W)
    xr = (x.clamp(min=0) if relu_gath else x).double()
    gr = (g.clamp(min=0) if relu_rows else g).double()
    xp = F.pad(xr, (h, h, h, h), mode='reflect') if reflect else F.pad(xr, (h, h, h, h))
    ref = torch.stack([torch.einsum('nryx,ncyx->rc', gr, xp[:, :, h + dy:h + dy + H, h + dx:h + dx + W]) for dy, dx in taps], 1)
    xg = torch.zeros(N, H, W, cin, device=DEV)
    xg.copy_(x.permute(0, 2, 3, 1))
    gg = torch.full((N, H, W, 16), 7.0, device=DEV)       # channels behind rows_used must not reach rows < rows_used
    gg[..., :R] = g.permute(0, 2, 3, 1).to(DEV)
    dw = torch.zeros(16, len(taps) * cin, device=DEV)
    dy = (_i8 * len(taps))(*[t[0] for t in taps])
    dx = (_i8 * len(taps))(*[t[1] for t in taps])
    check(lib().sdn_conv_wgrad_narrow(ptr(gg), ptr(xg), ptr(dw), N, H, W, 16, R, H, W, cin, len(taps), dy, dx, int(reflect),
                                      int(relu_rows), int(relu_gath), stream()))
    torch.cuda.synchronize()
    got = dw[:R].reshape(R, len(taps), cin).double().cpu()
    return name, float((got - ref).abs().max()) / float(ref.abs().max())


@pytest.mark.parametrize('case', range(len(CASES)))
def test_narrow_weight_gradient_matches_float64(case):
    name, err = _run(case)
    assert err <= 2e-6, (name, err)      # fp32 sums of <= 3k products per workgroup, atomics between workgroups


def test_the_column_kernel_still_agrees_on_dense_windows():
    """SDN_WGRAD_NARROW_ROW=0 (read once per process) sends dense 7 x 7 windows to k_wgrad_narrow as before r05."""
    code = ('import sys; sys.path.insert(0, %r); import test_gpu_wgrad_narrow as t\n'
            'for c in (0, 1, 2, 3):\n'
            '    name, err = t._run(c)\n'
            '    assert err <= 2e-6, (name, err)\n'
            'print("column kernel ok")\n' % os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDN_WGRAD_NARROW_ROW='0')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'column kernel ok' in r.stdout, r.stdout[-800:] + r.stderr[-1600:]


def test_the_row_kernel_is_exact_beside_an_mfma_kernel_on_another_stream():
    """r06 regression.  The r05 row kernel wrote its packed FMAs as inline `v_pk_fma_f32 ... op_sel:[0,1,0]` (low result lane reading
    the high half of the input pair).  Alone on the chip that is exact -- every case above passed -- but beside a wave of an MFMA
    kernel on the same SIMD gfx950 returns wrong LOW results: with sdn_conv_head_mfma (or sdn_conv_gemm) running on another
    stream, every even output row of the weight gradient was off by 1e-4 ... 2e-3 and every odd row exact.  That is exactly the
    product's schedule (weight gradients on a side stream beside the data-gradient chain).  Here: the generator-head shape at
    192 x 624, batch 4, four output rows, the head kernel's 64-row data-gradient launch on the main stream meanwhile -- on a
    private copy of d(out), so no buffer is shared -- four rounds; every row within 2e-6 of float64."""
    import torch.nn as nn
    from sdn_hip import check, lib, ptr
    from sdn_hip import conv as hc
    from sdn_hip import convplan as cp
    N, H, W, C, R = 4, 192, 624, 64, 4
    torch.manual_seed(77)
    x = torch.randn(N, H, W, C, device=DEV)
    dz = torch.zeros(N, H, W, 16, device=DEV)
    dz[..., :R] = torch.randn(N, H, W, R, device=DEV)
    dz2 = dz.clone()
    dy = (_i8 * 49)(*[k // 7 - 3 for k in range(49)])
    dx = (_i8 * 49)(*[k % 7 - 3 for k in range(49)])
    xp = F.relu(F.pad(x.permute(0, 3, 1, 2).double().cpu(), (3, 3, 3, 3), mode='reflect'))
    dzc = dz[..., :R].double().cpu()
    ref = torch.zeros(R, 49, C, dtype=torch.float64)
    for t in range(49):
        ky, kx = t // 7, t % 7
        ref[:, t] = torch.einsum('nhwr,nchw->rc', dzc, xp[:, :, ky:ky + H, kx:kx + W])
    conv = nn.Conv2d(64, R, 7, padding=3).to(DEV)
    st = hc.Stage('conv', conv, 0, reflect=3)
    launches, (GH, GW) = cp.conv_dgrad(7, 1, 3, H, W, True)
    L = launches[0]
    e = st.head_mfma('dgrad', L.taps, L.tapidx, 16, None)
    e.refresh()
    KH, KW, dy_min, dx_min, RR = e.meta
    target = torch.empty(N, GH, GW, 64, device=DEV)
    side = torch.cuda.Stream()
    worst = 0.0
    for _ in range(4):
        dw = torch.zeros(16, 49 * C, device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            check(lib().sdn_conv_wgrad_narrow(ptr(dz), ptr(x), ptr(dw), N, H, W, 16, R, H, W, C, 49, dy, dx, 1, 0, 1,
                                              ctypes.c_void_p(side.cuda_stream)))
        main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            check(lib().sdn_conv_head_mfma(ptr(dz2), N, H, W, 16, ptr(target), GH, GW, 64, RR, ptr(e.buf), KH, KW, dy_min, dx_min,
                                           0, 0, None, 0, None, main))
        torch.cuda.synchronize()
        got = dw[:R].double().cpu().reshape(R, 49, C)
        worst = max([worst] + [float((got[r] - ref[r]).norm() / ref[r].norm()) for r in range(R)])
    assert worst <= 2e-6, worst
