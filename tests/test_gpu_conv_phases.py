"""sdn_conv_gemm_phases (r05): the s*s phase launches of a ConvTranspose2d forward / a strided Conv2d's data gradient
(textural/models/networks.py:224-233, 297-303, 420-433 and their autograd) as ONE launch -- through the C ABI against (i) the
per-phase sdn_conv_gemm calls it replaces (same tiles, same K order: bit-equal outputs) and (ii) torch's float64 convolutions
on the CPU.  Shapes: 3x3 stride-2 transposed conv (1 + 2 + 2 + 4 taps), the 4x4 stride-2 discriminator layers' data gradient on
odd grids (phases of different sizes), a 7x7 stride-2 data gradient (16-tap phase), 32- / 64- / 128-channel N tiles, bias +
statistics, accumulate."""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
_i8, _i32 = ctypes.c_int8, ctypes.c_int32

CASES = [  # name, kind, N, IH, IW, cin, cout, k, s, p, extras       (IH x IW: the layer's INPUT grid)
    ('convT 3x3 s2 128 -> 64, bias + stats', 'convT', 2, 9, 14, 128, 64, 3, 2, 1, 'bias,stats'),
    ('convT 3x3 s2 64 -> 160 (128-wide N tiles, ragged)', 'convT', 1, 12, 10, 64, 160, 3, 2, 1, 'bias'),
    ('dgrad of 4x4 s2 pad 2, odd grid', 'dgrad', 2, 25, 31, 32, 64, 4, 2, 2, ''),
    ('dgrad of 3x3 s2 pad 1, accumulate', 'dgrad', 2, 24, 36, 64, 128, 3, 2, 1, 'acc'),
    ('dgrad of 7x7 s2 pad 3 (16-tap phase), 16 -> 32 channels', 'dgrad', 1, 20, 22, 16, 48, 7, 2, 3, ''),
]


def _cl(t, cp):
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cp, device=DEV)
    out[..., :c] = t.to(DEV).permute(0, 2, 3, 1)
    return out.contiguous()


@pytest.mark.parametrize('case', range(len(CASES)))
def test_phases_launch_equals_the_per_phase_launches_and_float64(case):
    from sdn_hip import check, lib, ptr, stream
    from sdn_hip import convplan as cp
    name, kind, N, IH, IW, cin, cout, k, s, p, extras = CASES[case]
    torch.manual_seed(500 + case)
    L = lib()
    if kind == 'convT':
        x = torch.randn(N, cin, IH, IW)
        w = torch.randn(cin, cout, k, k) * 0.1
        bias = torch.randn(cout) if 'bias' in extras else None
        ref = F.conv_transpose2d(x.double(), w.double(), None if bias is None else bias.double(), stride=s, padding=p, output_padding=1)
        launches, (OH, OW) = cp.convT_fwd(k, s, p, 1, IH, IW)
        R, C, sr, sc = cout, cin, k * k, cout * k * k          # rows = cout, columns = cin of the [cin, cout, k, k] weight
        gin, GH, GW, cg_in, cg_out = x, IH, IW, cin, cout
    else:
        # data gradient of y = conv2d(x [N, cin, IH, IW], w [cout, cin, k, k], stride s, pad p): gathers d(y), writes d(x)
        OHf, OWf = (IH + 2 * p - k) // s + 1, (IW + 2 * p - k) // s + 1
        dy_t = torch.randn(N, cout, OHf, OWf)
        w = torch.randn(cout, cin, k, k) * 0.1
        bias = None
        xx = torch.zeros(N, cin, IH, IW, dtype=torch.float64, requires_grad=True)
        F.conv2d(xx, w.double(), None, stride=s, padding=p).backward(dy_t.double())
        ref = xx.grad
        launches, (OH, OW) = cp.conv_dgrad(k, s, p, IH, IW, False)
        launches = [Lh for Lh in launches if Lh.taps]
        R, C, sr, sc = cin, cout, k * k, cin * k * k            # rows = cin (what is written), columns = cout (what is gathered)
        gin, GH, GW, cg_in, cg_out = dy_t, OHf, OWf, cout, cin
    assert tuple(ref.shape[2:]) == (OH, OW) and 2 <= len(launches) <= 4
    Cip, Cop = cp.cpad(cg_in), cp.cpad(cg_out)
    xg = _cl(gin, Cip)
    wg = w.to(DEV)
    bg = None
    if bias is not None:
        bg = torch.zeros(Cop, device=DEV)
        bg[:cg_out] = bias.to(DEV)
    rows = cp.weight_rows(Cop)
    packed, kps = [], []
    for Lh in launches:
        nt = len(Lh.taps)
        Kp = cp.kpad(nt, Cip)
        tix = torch.tensor(list(Lh.tapidx), dtype=torch.int32, device=DEV)
        buf = torch.empty(2 * rows * Kp, dtype=torch.bfloat16, device=DEV)
        check(L.sdn_conv_pack_weights(ptr(wg), R, C, sr, sc, ptr(tix), nt, Cip, Kp, rows, ptr(buf), stream()))
        packed.append(buf)
        kps.append(Kp)
    base = torch.randn(N, OH, OW, Cop, device=DEV) if 'acc' in extras else None
    want_stats = 'stats' in extras

    def fresh():
        out = base.clone() if base is not None else torch.full((N, OH, OW, Cop), float('nan'), device=DEV)
        st = torch.zeros(N, 8, Cop, 2, dtype=torch.float64, device=DEV) if want_stats else None
        return out, st
    # (i) one sdn_conv_gemm call per phase
    out_a, st_a = fresh()
    for Lh, buf, Kp in zip(launches, packed, kps):
        nt = len(Lh.taps)
        dy = (_i8 * nt)(*[t[0] for t in Lh.taps])
        dx = (_i8 * nt)(*[t[1] for t in Lh.taps])
        check(L.sdn_conv_gemm(ptr(xg), N, GH, GW, Cip, ptr(out_a), OH, OW, Cop, Lh.QH, Lh.QW, Lh.istride, Lh.ostride, Lh.py, Lh.px,
                              nt, dy, dx, 0, 0, ptr(buf), Kp, rows, ptr(bg), 0, ptr(st_a), int(base is not None), 3, None, 0, stream()))
    # (ii) all phases in one launch
    out_b, st_b = fresh()
    n = len(launches)
    arr = lambda vals: (_i32 * n)(*vals)   # noqa: E731
    taps = []
    for Lh in launches:
        taps += [t[0] for t in Lh.taps] + [t[1] for t in Lh.taps]
    taps_c = (_i8 * len(taps))(*taps)
    wps = (ctypes.c_void_p * n)(*[b.data_ptr() for b in packed])
    check(L.sdn_conv_gemm_phases(ptr(xg), N, GH, GW, Cip, ptr(out_b), OH, OW, Cop, launches[0].istride, launches[0].ostride, n,
                                 arr([Lh.QH for Lh in launches]), arr([Lh.QW for Lh in launches]), arr([Lh.py for Lh in launches]),
                                 arr([Lh.px for Lh in launches]), arr([len(Lh.taps) for Lh in launches]), taps_c, 0, 0, wps,
                                 arr(kps), rows, ptr(bg), 0, ptr(st_b), int(base is not None), 3, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b), (name, 'phases launch differs from the per-phase launches in %d elements'
                                       % int((out_a != out_b).sum()))
    got = out_b[..., :cg_out].permute(0, 3, 1, 2).double().cpu()
    want = ref if base is None else ref + base[..., :cg_out].permute(0, 3, 1, 2).double().cpu()
    err = float((got - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-5, (name, err)
    if Cop > cg_out and base is None:
        assert float(out_b[..., cg_out:].abs().max()) == 0.0
    if want_stats:
        a, b_ = st_a.sum(1), st_b.sum(1)
        assert float((a - b_).abs().max()) <= 1e-9 * float(a.abs().max())
        pre = ref
        s1 = b_[:, :cg_out, 0].cpu()
        r1, r2 = pre.sum((2, 3)), (pre * pre).sum((2, 3))
        assert float((s1 - r1).abs().max()) <= 1e-5 * float(r1.abs().max() + r2.sqrt().max())


def test_phases_argument_errors():
    from sdn_hip import SdnHipError, check, lib, ptr, stream
    x = torch.zeros(1, 4, 4, 16, device=DEV)
    out = torch.zeros(1, 8, 8, 32, device=DEV)
    w = torch.zeros(2 * 32 * 32, dtype=torch.bfloat16, device=DEV)
    one = (_i32 * 1)(4)
    zero = (_i32 * 1)(0)
    taps = (_i8 * 34)(*([0] * 34))
    wps = (ctypes.c_void_p * 1)(w.data_ptr())
    with pytest.raises(SdnHipError):   # 17 taps in a phase
        check(lib().sdn_conv_gemm_phases(ptr(x), 1, 4, 4, 16, ptr(out), 8, 8, 32, 1, 2, 1, one, one, zero, zero, (_i32 * 1)(17),
                                         taps, 0, 0, wps, (_i32 * 1)(288), 32, None, 0, None, 0, 3, stream()))
    with pytest.raises(SdnHipError):   # the phase's grid leaves the output tensor
        check(lib().sdn_conv_gemm_phases(ptr(x), 1, 4, 4, 16, ptr(out), 8, 8, 32, 1, 2, 1, (_i32 * 1)(5), one, zero, zero,
                                         (_i32 * 1)(1), taps, 0, 0, wps, (_i32 * 1)(32), 32, None, 0, None, 0, 3, stream()))
