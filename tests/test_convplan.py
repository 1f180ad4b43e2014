"""CPU tests of the conv launch geometry (sdn_hip/convplan.py): every layer type of the textural networks --
forward, data gradient and weight gradient -- executed by the numpy stand-in for the kernels (tests/conv_emul.py) and
compared with torch.nn.functional / autograd in float64."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import conv_emul as em  # noqa: E402
from sdn_hip import convplan as cp  # noqa: E402

# (kind, k, s, p, reflect, op, cin, cout, H, W): the layer shapes of networks.py:211-239, 286-308, 412-449
CASES = [
    ('conv', 7, 1, 3, True, 0, 3, 4, 9, 11),    # c7s1 stem / head behind ReflectionPad2d(3)
    ('conv', 3, 2, 1, False, 0, 4, 6, 8, 10),   # downsampling conv
    ('conv', 3, 2, 1, False, 0, 4, 6, 7, 9),    # ... odd sizes
    ('conv', 3, 1, 1, True, 0, 5, 5, 6, 7),     # ResnetBlock conv behind ReflectionPad2d(1)
    ('conv', 3, 1, 1, False, 0, 3, 5, 6, 7),    # zero padded stride 1
    ('conv', 4, 2, 2, False, 0, 3, 4, 8, 11),   # discriminator 4x4 stride 2 pad 2
    ('conv', 4, 1, 2, False, 0, 4, 2, 5, 6),    # discriminator 4x4 stride 1 pad 2
    ('convT', 3, 2, 1, False, 1, 6, 4, 4, 5),   # upsampling ConvTranspose2d
]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2)


def torch_fwd(kind, x, w, k, s, p, reflect, op):
    if kind == 'conv':
        if reflect:
            return F.conv2d(F.pad(x, (p, p, p, p), mode='reflect'), w, None, s, 0)
        return F.conv2d(x, w, None, s, p)
    return F.conv_transpose2d(x, w, None, s, p, op)


def strides(kind, cin, cout, k):
    kk = k * k
    if kind == 'conv':
        return (cin * kk, kk), (kk, cin * kk)
    return (kk, cout * kk), (cout * kk, kk)


@pytest.mark.parametrize('case', CASES, ids=lambda c: '%s_k%d_s%d_p%d_%s' % (c[0], c[1], c[2], c[3], 'refl' if c[4] else 'zero'))
def test_forward_dgrad_wgrad(case):
    kind, k, s, p, reflect, op, cin, cout, H, W = case
    g = torch.Generator().manual_seed(7)
    N = 2
    x = torch.randn(N, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = torch.randn(*wshape, generator=g, dtype=torch.float64, requires_grad=True)
    y = torch_fwd(kind, x, w, k, s, p, reflect, op)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    cip, cop = cp.cpad(cin), cp.cpad(cout)
    xa = np.zeros((N, H, W, cip))
    xa[..., :cin] = nhwc(x.detach())
    wf = w.detach().numpy().reshape(-1)
    (sr_f, sc_f), (sr_d, sc_d) = strides(kind, cin, cout, k)
    pad_mode = 1 if reflect else 0

    # ---- forward
    if kind == 'conv':
        launches, (OH, OW) = cp.conv_fwd(k, s, p, H, W)
    else:
        launches, (OH, OW) = cp.convT_fwd(k, s, p, op, H, W)
    assert (OH, OW) == tuple(y.shape[2:])
    out = np.full((N, OH, OW, cop), np.nan)
    for L in launches:
        pk = em.pack(wf, cout, cin, sr_f, sc_f, L.tapidx, cip, cp.weight_rows(cop))
        em.gemm(xa, out, L, pad_mode, pk)
    assert not np.isnan(out).any(), 'the phases must cover every output position'
    np.testing.assert_allclose(out[..., :cout], nhwc(y.detach()), rtol=1e-10, atol=1e-10)
    assert np.all(out[..., cout:] == 0)

    # ---- data gradient
    dz = np.zeros((N, OH, OW, cop))
    dz[..., :cout] = nhwc(gy)
    if kind == 'conv':
        launches, (GH, GW) = cp.conv_dgrad(k, s, p, H, W, reflect)
    else:
        launches, (GH, GW) = cp.convT_dgrad(k, s, p, H, W)
    gx = np.full((N, GH, GW, cip), np.nan)
    for L in launches:
        pk = em.pack(wf, cin, cout, sr_d, sc_d, L.tapidx, cop, cp.weight_rows(cip))
        em.gemm(dz, gx, L, 0, pk)
    assert not np.isnan(gx).any()
    if reflect:
        gx = em.reflect_fold(gx, H, W, p)
    np.testing.assert_allclose(gx[..., :cin], nhwc(x.grad), rtol=1e-10, atol=1e-10)

    # ---- weight gradient
    gw = np.zeros(wf.shape)
    if kind == 'conv':
        WL = cp.conv_wgrad(k, s, p, OH, OW)
        dw = em.wgrad(dz, xa, WL, pad_mode)
        em.unpack(dw, cout, cin, sr_f, sc_f, WL.tapidx, cip, gw)
    else:
        WL = cp.convT_wgrad(k, s, p, H, W)
        dw = em.wgrad(xa, dz, WL, 0)
        em.unpack(dw, cin, cout, sr_d, sc_d, WL.tapidx, cop, gw)
    np.testing.assert_allclose(gw.reshape(wshape), w.grad.numpy(), rtol=1e-10, atol=1e-10)


def test_deferred_relu_matches_materialised():
    """in_relu on load == conv of relu(x); its gradient masks by x > 0 (done by the producer's backward)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 6, 6, generator=g, dtype=torch.float64)
    w = torch.randn(4, 3, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(F.relu(x), w, None, 1, 1)
    launches, (OH, OW) = cp.conv_fwd(3, 1, 1, 6, 6)
    xa = np.zeros((1, 6, 6, 16))
    xa[..., :3] = nhwc(x)
    out = np.zeros((1, OH, OW, 16))
    pk = em.pack(w.numpy().reshape(-1), 4, 3, 27, 9, launches[0].tapidx, 16, 32)
    em.gemm(xa, out, launches[0], 0, pk, in_relu=True)
    np.testing.assert_allclose(out[..., :4], nhwc(y), rtol=1e-12, atol=1e-12)


def test_padding_helpers():
    assert cp.cpad(3) == 16 and cp.cpad(48) == 48 and cp.cpad(18) == 32
    assert cp.cpad_pow2(48) == 64 and cp.cpad_pow2(1) == 16 and cp.cpad_pow2(1024) == 1024
    assert cp.weight_rows(16) == 32 and cp.weight_rows(48) == 64 and cp.weight_rows(64) == 64
    assert cp.weight_rows(128) == 128 and cp.weight_rows(192) == 256
    assert cp.kpad(49, 48) % 32 == 0 and cp.kpad(49, 48) >= 49 * 48
    assert cp.wgrad_splits(4 * 384 * 1248, 19) > 1 and cp.wgrad_splits(100, 4) == 1


# ---------------------------------------------------------------------------------------------------- narrow layers
def _narrow_emul(x, dense, KH, KW, dy_min, dx_min, reflect, OH, OW):
    """numpy stand-in for sdn_conv_narrow_fwd: out[n, y, x, r] = sum_{dy,dx,c} f(in[n, y+dy_min+dy, x+dx_min+dx, c]) W[dy,dx,c,r]"""
    N, IH, IW, C = x.shape
    out = np.zeros((N, OH, OW, dense.shape[3]))
    for dyi in range(KH):
        for dxi in range(KW):
            ys = np.arange(OH) + dy_min + dyi
            xs = np.arange(OW) + dx_min + dxi
            if reflect:
                ys, xs = np.abs(ys), np.abs(xs)
                ys, xs = np.minimum(ys, 2 * IH - 2 - ys), np.minimum(xs, 2 * IW - 2 - xs)
            oky, okx = (ys >= 0) & (ys < IH), (xs >= 0) & (xs < IW)
            patch = x[:, np.clip(ys, 0, IH - 1)][:, :, np.clip(xs, 0, IW - 1)] * (oky[:, None] & okx[None, :])[None, :, :, None]
            out += patch @ dense[dyi, dxi]
    return out


@pytest.mark.parametrize('k,p,reflect,cin,cout', [(7, 3, True, 20, 3), (4, 2, False, 32, 1), (3, 1, False, 16, 5)])
def test_narrow_dense_weights_forward_and_restricted_dgrad(k, p, reflect, cin, cout):
    """Stage.narrow (the dense tap window sdn_conv_narrow_fwd reads) reproduces the layer forward, and -- in the
    'dgrad' orientation with a row range -- the matching channel slice of the input gradient."""
    from sdn_hip import conv as hc
    g = torch.Generator().manual_seed(k * 10 + cout)
    conv = torch.nn.Conv2d(cin, cout, k, 1, 0 if reflect else p)   # float32 parameters, as in the product
    st = hc.Stage('conv', conv, 0, reflect=p if reflect else 0)
    H, W, N = 9, 11, 2
    x = torch.randn(N, cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    y = torch_fwd('conv', x, conv.weight.detach().double(), k, 1, p, reflect, 0)
    cip = cp.cpad(cin)
    xa = np.zeros((N, H, W, cip))
    xa[..., :cin] = nhwc(x.detach())
    launches, (OH, OW) = cp.conv_fwd(k, 1, p, H, W)
    e = st.narrow('fwd', launches[0].taps, launches[0].tapidx, cip)   # a persistent buffer + the torch ops refreshing it
    e.refresh()
    dense, (KH, KW, dy_min, dx_min, R) = e.buf, e.meta
    assert (KH, KW, R) == (k, k, cout) and dense.shape == (k, k, cip, 1 if cout == 1 else (4 if cout <= 4 else 8))
    out = _narrow_emul(xa, dense.numpy(), KH, KW, dy_min, dx_min, reflect, OH, OW)
    np.testing.assert_allclose(out[..., :cout], nhwc(y.detach()), rtol=1e-10, atol=1e-10)
    assert np.all(out[..., cout:] == 0)
    # data gradient restricted to input channels [lo, hi): narrow 'dgrad' over d(out) (zero outside), padded grid + fold
    if cin >= 9:
        lo, hi = 4, 9
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(gy)
        cop = cp.cpad(cout)
        dz = np.zeros((N, OH, OW, cop))
        dz[..., :cout] = nhwc(gy)
        launches, (GH, GW) = cp.conv_dgrad(k, 1, p, H, W, reflect)
        e = st.narrow('dgrad', launches[0].taps, launches[0].tapidx, cop, (lo, hi))
        e.refresh()
        dense, (KH, KW, dy_min, dx_min, R) = e.buf, e.meta
        assert R == hi - lo
        gx = _narrow_emul(dz, dense.numpy(), KH, KW, dy_min, dx_min, False, GH, GW)
        if reflect:
            gx = em.reflect_fold(gx, H, W, p)
        np.testing.assert_allclose(gx[..., :hi - lo], nhwc(x.grad)[..., lo:hi], rtol=1e-10, atol=1e-10)
