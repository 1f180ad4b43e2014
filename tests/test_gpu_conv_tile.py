"""The r04 tiled MFMA kernels through the C ABI -- sdn_split_planes, sdn_conv_pack_weights_kmajor, sdn_conv_tile, sdn_conv_halo,
sdn_conv_wgrad_tile (include/sdn_hip.h) -- against torch's float64 convolutions on the CPU (the layer arithmetic the pinned
oracle/textural_oracle.py is made of: textural/models/networks.py:211-283, 412-461), on shapes that reach every branch: reflect /
zero padding, stride 2, transposed-convolution phases, bias / LeakyReLU / InstanceNorm statistics / accumulate epilogues, ragged
tiles (positions, channels, K slices), both row-tile sizes of the weight gradient, windows the patch kernel takes and refuses.
Gate: 1e-5 of the output scale (bf16 x 3 products, fp32 accumulation: ~2e-6 measured), statistics 1e-5 relative."""
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
_i8 = ctypes.c_int8


def _planes(x, relu=False):
    from sdn_hip import check, lib, ptr, stream
    n = x.numel()
    stride = (n + 7) // 8 * 8
    pl = torch.empty(2 * stride, dtype=torch.bfloat16, device=x.device)
    check(lib().sdn_split_planes(ptr(x), n, int(relu), ptr(pl), stride, stream()))
    return pl, stride


def test_split_planes_reconstruct_the_input():
    x = torch.randn(3, 7, 5, 16, device=DEV) * 3
    pl, stride = _planes(x)
    hi, lo = pl[:x.numel()].float().reshape(x.shape), pl[stride:stride + x.numel()].float().reshape(x.shape)
    assert torch.equal(hi, x.bfloat16().float())
    assert float((hi + lo - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    plr, _ = _planes(x, relu=True)
    assert torch.equal(plr[:x.numel()].float().reshape(x.shape), x.clamp(min=0).bfloat16().float())


def _cl(t, cp):
    """NCHW -> channels-last fp32 padded to cp channels"""
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cp, device=DEV)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out.contiguous()


CONV_CASES = [  # name, kind, N, IH, IW, cin, cout, k, s, p, reflect, extras
    ('3x3 reflect + stats', 'fwd', 2, 20, 30, 64, 128, 3, 1, 1, 1, 'bias,stats'),
    ('3x3 zero pad, lrelu, ragged channels', 'fwd', 2, 21, 33, 64, 160, 3, 1, 1, 0, 'bias,lrelu'),
    ('4x4 s1 pad 2', 'fwd', 1, 13, 19, 96, 160, 4, 1, 2, 0, 'bias'),
    ('4x4 s2 pad 2, 64-channel tile', 'fwd', 2, 25, 31, 32, 64, 4, 2, 2, 0, 'bias,lrelu,stats'),
    ('3x3 s2', 'fwd', 2, 24, 36, 64, 128, 3, 2, 1, 0, 'stats'),
    ('convT 3x3 s2 phases', 'convT', 2, 9, 14, 64, 96, 3, 2, 1, 0, 'bias'),
    ('3x3 accumulate', 'fwd', 1, 17, 22, 128, 64, 3, 1, 1, 1, 'acc'),
    ('1 position tile, 7x7', 'fwd', 1, 9, 9, 32, 48, 7, 1, 3, 1, 'bias'),
    # K-split tails (the data-gradient grids of the residual layers): 288 positions = one full tile + 32 rows per image
    ('K-split tail, 4 even slices', 'fwd', 2, 18, 16, 64, 160, 3, 1, 1, 0, 'bias ksplit4'),
    ('K-split tail, 27 steps in 4 slices, reflect', 'fwd', 2, 18, 16, 96, 128, 3, 1, 1, 1, 'bias ksplit4'),
    ('K-split, the only tile of the image', 'fwd', 1, 9, 9, 64, 48, 3, 1, 1, 0, 'ksplit3'),
]


@pytest.mark.parametrize('case', range(len(CONV_CASES)))
def test_conv_tile_and_halo_match_float64(case):
    from sdn_hip import check, lib, ptr, stream
    from sdn_hip import convplan as cp
    name, kind, N, IH, IW, cin, cout, k, s, p, reflect, extras = CONV_CASES[case]
    torch.manual_seed(100 + case)
    x = torch.randn(N, cin, IH, IW)
    bias = torch.randn(cout) if 'bias' in extras else None
    if kind == 'fwd':
        w = torch.randn(cout, cin, k, k) * 0.1
        xp = F.pad(x.double(), (p, p, p, p), mode='reflect') if reflect else x.double()
        ref = F.conv2d(xp, w.double(), None if bias is None else bias.double(), stride=s, padding=0 if reflect else p)
        launches, (OH, OW) = cp.conv_fwd(k, s, p, IH, IW)
        R, C, sr, sc = cout, cin, cin * k * k, k * k
    else:
        w = torch.randn(cin, cout, k, k) * 0.1
        ref = F.conv_transpose2d(x.double(), w.double(), None if bias is None else bias.double(), stride=s, padding=p, output_padding=1)
        launches, (OH, OW) = cp.convT_fwd(k, s, p, 1, IH, IW)
        R, C, sr, sc = cout, cin, k * k, cout * k * k
    assert tuple(ref.shape[2:]) == (OH, OW)
    Cip, Cop = cp.cpad(cin), cp.cpad_pow2(cout) if cout not in (96, 160, 48) else cp.cpad(cout)
    Cip = (Cip + 31) // 32 * 32
    xg = _cl(x.to(DEV), Cip)
    wg = w.to(DEV)
    bg = None
    if bias is not None:
        bg = torch.zeros(Cop, device=DEV)
        bg[:cout] = bias.to(DEV)
    act = 1 if 'lrelu' in extras else 0
    ksplit = int(extras.split('ksplit')[1]) if 'ksplit' in extras else 0
    pre = ref.clone()
    if act:
        ref = F.leaky_relu(ref, 0.2)
    base = torch.randn(N, OH, OW, Cop, device=DEV) if 'acc' in extras else None
    pl, pstride = _planes(xg)
    L = lib()
    rows = cp.tile_weight_rows(Cop)

    def run(entry):
        out = base.clone() if base is not None else torch.full((N, OH, OW, Cop), float('nan'), device=DEV)
        st = torch.zeros(N, 8, Cop, 2, dtype=torch.float64, device=DEV) if 'stats' in extras else None
        for Lh in launches:
            nt = len(Lh.taps)
            tix = torch.tensor(list(Lh.tapidx), dtype=torch.int32, device=DEV)
            packed = torch.empty(2 * rows * nt * Cip, dtype=torch.bfloat16, device=DEV)
            check(L.sdn_conv_pack_weights_kmajor(ptr(wg), R, C, sr, sc, ptr(tix), nt, Cip, rows, ptr(packed), stream()))
            dy = (_i8 * nt)(*[t[0] for t in Lh.taps])
            dx = (_i8 * nt)(*[t[1] for t in Lh.taps])
            if entry == 'tile':
                check(L.sdn_conv_tile(ptr(pl), pstride, N, IH, IW, Cip, ptr(out), None, 0, 0, OH, OW, Cop, Lh.QH, Lh.QW, Lh.istride,
                                      Lh.ostride, Lh.py, Lh.px, nt, dy, dx, reflect, ptr(packed), rows, ptr(bg), act, ptr(st),
                                      int(base is not None), ksplit, stream()))
            else:
                check(L.sdn_conv_halo(ptr(pl), pstride, N, IH, IW, Cip, ptr(out), OH, OW, Cop, nt, dy, dx, reflect, ptr(packed), rows,
                                      ptr(bg), act, ptr(st), int(base is not None), stream()))
        torch.cuda.synchronize()
        return out, st
    entries = ['tile']
    halo = (len(launches) == 1 and s == 1 and kind == 'fwd' and k * k >= 9 and Cop > 64 and not ksplit)
    if halo:
        entries.append('halo')
    for entry in entries:
        out, st = run(entry)
        got = out[..., :cout].permute(0, 3, 1, 2).double().cpu()
        want = ref if base is None else ref + base[..., :cout].permute(0, 3, 1, 2).double().cpu()
        err = float((got - want).abs().max()) / float(want.abs().max())
        assert err <= 1e-5, (name, entry, err)
        if Cop > cout and base is None:
            assert float(out[..., cout:].abs().max()) == 0.0, (name, entry, 'padding channels must come out as zeros')
        if st is not None:
            s1 = st.sum(1)[:, :cout, 0].cpu()
            s2 = st.sum(1)[:, :cout, 1].cpu()
            r1, r2 = pre.sum((2, 3)), (pre * pre).sum((2, 3))
            assert float((s1 - r1).abs().max()) <= 1e-5 * float(r1.abs().max() + r2.sqrt().max()), (name, entry, 'sum')
            assert float(((s2 - r2).abs() / r2).max()) <= 1e-5, (name, entry, 'sum of squares')


def test_halo_refuses_what_it_cannot_take():
    from sdn_hip import SdnHipError, check, lib, ptr, stream
    x = torch.zeros(1, 8, 8, 32, device=DEV)
    pl, ps = _planes(x)
    w = torch.zeros(2 * 128 * 4 * 32, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(1, 8, 8, 128, device=DEV)
    dy = (_i8 * 4)(0, 0, 1, 1)
    dx = (_i8 * 4)(0, 1, 0, 1)
    with pytest.raises(SdnHipError):   # a 2x2 window: too few taps for the patch pipeline
        check(lib().sdn_conv_halo(ptr(pl), ps, 1, 8, 8, 32, ptr(out), 8, 8, 128, 4, dy, dx, 0, ptr(w), 128, None, 0, None, 0, stream()))
    dy9 = (_i8 * 9)(0, 0, 0, 1, 1, 1, 2, 2, 3)
    dx9 = (_i8 * 9)(0, 1, 2, 0, 1, 2, 0, 1, 0)
    with pytest.raises(SdnHipError):   # nine taps that do not fill their window
        check(lib().sdn_conv_halo(ptr(pl), ps, 1, 8, 8, 32, ptr(out), 8, 8, 128, 9, dy9, dx9, 0, ptr(w), 128, None, 0, None, 0, stream()))


WGRAD_CASES = [  # name, kind, N, OH, OW, cout, cin, k, s, p, reflect
    ('3x3 reflect, 128-row tiles', 'conv', 2, 12, 18, 128, 64, 3, 1, 1, 1),
    ('3x3 s2, 64-row tiles, 48 gathered channels', 'conv', 2, 11, 17, 64, 48, 3, 2, 1, 0),
    ('4x4 s1 pad 2, ragged rows', 'conv', 1, 14, 20, 160, 32, 4, 1, 2, 0),
    ('7x7 reflect, 49 taps', 'conv', 1, 16, 16, 64, 16, 7, 1, 3, 1),
    ('convT 3x3 s2', 'convT', 2, 9, 13, 96, 128, 3, 2, 1, 0),
    ('a handful of positions', 'conv', 1, 3, 5, 64, 32, 3, 1, 1, 0),
]


@pytest.mark.parametrize('case', range(len(WGRAD_CASES)))
@pytest.mark.parametrize('mode', ['0', '1'])
def test_wgrad_tile_matches_float64(case, mode, monkeypatch):
    """both work decompositions (SDN_WTILE_MODE: 0 stream-K ranges, 1 K slices where they fit)"""
    from sdn_hip import check, lib, ptr, stream
    from sdn_hip import convplan as cp
    monkeypatch.setenv('SDN_WTILE_MODE', mode)
    name, kind, N, OH, OW, cout, cin, k, s, p, reflect = WGRAD_CASES[case]
    torch.manual_seed(200 + case)
    if kind == 'conv':
        IH, IW = (OH - 1) * s + k - 2 * p, (OW - 1) * s + k - 2 * p
        x = torch.randn(N, cin, IH, IW)
        dz = torch.randn(N, cout, OH, OW)
        xd = x.double().requires_grad_(False)
        wd = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
        xp = F.pad(xd, (p, p, p, p), mode='reflect') if reflect else xd
        y = F.conv2d(xp, wd, None, stride=s, padding=0 if reflect else p)
        (y * dz.double()).sum().backward()
        ref = wd.grad.permute(0, 2, 3, 1).reshape(cout, k * k * cin)                    # [r][tap][c]
        Cr, Cc = cp.cpad(cout), cp.cpad(cin)
        rows, gath = _cl(dz.to(DEV), Cr), _cl(x.to(DEV), Cc)
        WL = cp.conv_wgrad(k, s, p, OH, OW)
        GH, GW, nr, nc = IH, IW, cout, cin
    else:
        GH, GW = cp.convT_out_size(OH, k, s, p, 1), cp.convT_out_size(OW, k, s, p, 1)
        x = torch.randn(N, cin, OH, OW)
        dz = torch.randn(N, cout, GH, GW)
        wd = torch.zeros(cin, cout, k, k, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose2d(x.double(), wd, None, stride=s, padding=p, output_padding=1)
        (y * dz.double()).sum().backward()
        ref = wd.grad.permute(0, 2, 3, 1).reshape(cin, k * k * cout)                    # rows = cin, cols = (tap, cout)
        Cr, Cc = cp.cpad(cin), cp.cpad(cout)
        rows, gath = _cl(x.to(DEV), Cr), _cl(dz.to(DEV), Cc)
        WL = cp.convT_wgrad(k, s, p, OH, OW)
        nr, nc = cin, cout
    nt = len(WL.taps)
    dy = (_i8 * nt)(*[t[0] for t in WL.taps])
    dx = (_i8 * nt)(*[t[1] for t in WL.taps])
    rp, rs = _planes(rows)
    gp, gs = _planes(gath)
    dw = torch.zeros(Cr, nt * Cc, device=DEV)
    check(lib().sdn_conv_wgrad_tile(ptr(rp), rs, ptr(gp), gs, ptr(dw), N, WL.QH, WL.QW, Cr, GH, GW, Cc, WL.istride, nt, dy, dx,
                                    reflect, stream()))
    torch.cuda.synchronize()
    got = dw.reshape(Cr, nt, Cc)[:nr, :, :nc].double().cpu()
    # the plan's tap order: tapidx[t] = ky * k + kx of tap t
    want = ref.reshape(nr, k * k, nc)[:, list(WL.tapidx), :]
    err = float((got - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-5, (name, err)
    assert float(dw.reshape(Cr, nt, Cc)[nr:].abs().max() if Cr > nr else 0.0) == 0.0


@pytest.mark.parametrize('shape', [(2, 37, 70, (14, 25, 1, 5, 3)), (1, 8, 64, (3,)), (3, 5, 129, (15, 3)), (1, 3, 9, (7, 1, 1, 1, 1, 1, 1, 1)),
                                   (70, 1000, 8, (3,))])   # N * H = 70 000 rows: beyond a grid.y
def test_assemble_nhwc_equals_cat_permute_pad(shape):
    """sdn_assemble_nhwc: the chain-input buffer torch.cat + permute + zero padding would build, bit for bit (ragged widths,
    one to eight parts, pad channels zero even when the destination held garbage)."""
    import ctypes

    from sdn_hip import check, lib, stream
    from sdn_hip import conv as hc
    from sdn_hip import convplan as cp
    N, H, W, chans = shape
    torch.manual_seed(sum(chans) + W)
    parts = [torch.randn(N, c, H, W, device=DEV) for c in chans]
    C = sum(chans)
    Cp = cp.cpad(C)
    want = torch.zeros(N, H, W, Cp, device=DEV)
    want[..., :C] = torch.cat(parts, dim=1).permute(0, 2, 3, 1)
    got = torch.full((N, H, W, Cp), float('nan'), device=DEV)
    ptrs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
    ch = (ctypes.c_int32 * len(parts))(*chans)
    check(lib().sdn_assemble_nhwc(ptrs, ch, len(parts), N, H, W, Cp, got.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(hc._input_buffer(parts), want)      # the executor takes the same path
