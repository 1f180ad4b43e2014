"""sdn_conv_head_mfma (r05, csrc/conv_head.hip): the 7 x 7 head layers of the generator / encoder
(textural/models/networks.py:236 -- ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh --, :306 -- 16 -> 5) and the stem's data gradient
towards the encoder features, on v_mfma_f32_16x16x32_bf16 with the output channels as matrix rows.  Through the C ABI against
torch's float64 convolution on the CPU (gate 5e-5 of the output scale: bf16 x 3 products, fp32 accumulation -- 1.6e-5 / 2.4e-5
measured behind the tanh, whose output scale is 1 whatever the pre-activation's) and against the
exact-fp32 vector kernel it replaces (sdn_conv_narrow_fwd), with the fragment-ordered weights built by the product's own
Stage.head_mfma; ragged grids (rows / columns that do not fill the 8 x 32 blocks), ReLU on load, zero and reflected borders."""
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

CASES = [  # name, N, H, W, cin, cout, reflect, in_relu, act
    ('generator head 64 -> 3, reflect, ReLU on load, tanh', 2, 24, 64, 64, 3, True, True, 'tanh'),
    ('encoder head 16 -> 5, reflect, tanh, ragged grid', 1, 21, 45, 16, 5, True, False, 'tanh'),
    ('64 -> 5, zero border, no activation (the stem data-gradient shape), ragged', 2, 13, 70, 64, 5, False, False, 'none'),
    ('one block, 16 -> 3', 1, 8, 32, 16, 3, True, False, 'none'),
]


def _cl(t, cp):
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cp, device=DEV)
    out[..., :c] = t.to(DEV).permute(0, 2, 3, 1)
    return out.contiguous()


@pytest.mark.parametrize('case', range(len(CASES)))
def test_head_mfma_matches_float64_and_the_fp32_vector_kernel(case):
    import ctypes

    import torch.nn as nn
    from sdn_hip import check, lib, ptr, stream
    from sdn_hip import conv as hc
    from sdn_hip import convplan as cp
    name, N, H, W, cin, cout, reflect, in_relu, act = CASES[case]
    torch.manual_seed(700 + case)
    conv = nn.Conv2d(cin, cout, 7, padding=0 if reflect else 3).to(DEV)
    with torch.no_grad():
        conv.weight.mul_(3.0)
    x = torch.randn(N, cin, H, W)
    xr = x.clamp(min=0) if in_relu else x
    xp = F.pad(xr.double(), (3, 3, 3, 3), mode='reflect') if reflect else xr.double()
    ref = F.conv2d(xp, conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=0 if reflect else 3)
    if act == 'tanh':
        ref = torch.tanh(ref)
    st = hc.Stage('conv', conv, 0, reflect=3 if reflect else 0)
    launches, (OH, OW) = cp.conv_fwd(7, 1, 3, H, W)
    Lh = launches[0]
    e = st.head_mfma('fwd', Lh.taps, Lh.tapidx, cin)
    e.refresh()
    nar = st.narrow('fwd', Lh.taps, Lh.tapidx, cin)
    KH, KW, dy_min, dx_min, R = e.meta
    steps = ctypes.c_int(0)
    check(lib().sdn_conv_head_steps(cin, KH, KW, ctypes.byref(steps)))
    assert (KH, KW, dy_min, dx_min) == (7, 7, -3, -3) and R >= cout and tuple(e.buf.shape) == (steps.value, 2, 64, 8)
    xg = _cl(x, cin)
    bias = torch.zeros(16, device=DEV)
    bias[:cout] = conv.bias.detach()
    a = {'none': 0, 'tanh': 2}[act]
    out = torch.full((N, OH, OW, 16), float('nan'), device=DEV)
    check(lib().sdn_conv_head_mfma(ptr(xg), N, H, W, cin, ptr(out), OH, OW, 16, cout, ptr(e.buf), KH, KW, dy_min, dx_min,
                                   int(reflect), int(in_relu), ptr(bias), a, None, stream()))
    out_n = torch.full((N, OH, OW, 16), float('nan'), device=DEV)
    check(lib().sdn_conv_narrow_fwd(ptr(xg), N, H, W, cin, ptr(out_n), OH, OW, 16, cout, ptr(nar.buf), KH, KW, dy_min, dx_min,
                                    int(reflect), int(in_relu), ptr(bias), a, stream()))
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).double().cpu()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err <= 5e-5, (name, err)
    assert float(out[..., cout:].abs().max()) == 0.0, 'channels behind the real ones must come out as zeros'
    err_n = float((out - out_n).abs().max()) / float(ref.abs().max())
    assert err_n <= 5e-5, (name, 'against sdn_conv_narrow_fwd', err_n)


def test_head_mfma_refuses_what_it_is_not_built_for():
    from sdn_hip import SdnHipError, check, lib, ptr, stream
    x = torch.zeros(1, 8, 32, 32, device=DEV)
    w = torch.zeros(200, 2, 64, 8, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(1, 8, 32, 16, device=DEV)
    with pytest.raises(SdnHipError):   # 32 input channels
        check(lib().sdn_conv_head_mfma(ptr(x), 1, 8, 32, 32, ptr(out), 8, 32, 16, 3, ptr(w), 7, 7, -3, -3, 1, 0, None, 0, None, stream()))
    x16 = torch.zeros(1, 8, 32, 16, device=DEV)
    with pytest.raises(SdnHipError):   # a 3 x 3 window
        check(lib().sdn_conv_head_mfma(ptr(x16), 1, 8, 32, 16, ptr(out), 8, 32, 16, 3, ptr(w), 3, 3, -1, -1, 0, 0, None, 0, None, stream()))
    out32 = torch.zeros(1, 8, 32, 32, device=DEV)
    with pytest.raises(SdnHipError):   # a 32-channel output tensor
        check(lib().sdn_conv_head_mfma(ptr(x16), 1, 8, 32, 16, ptr(out32), 8, 32, 32, 3, ptr(w), 7, 7, -3, -3, 0, 0, None, 0, None, stream()))


@pytest.mark.parametrize('case', [
    ('encoder stem 3 -> 16 under InstanceNorm: statistics epilogue, ragged grid', 2, 21, 45, 3, 16, 16, True),
    ('6 -> 8 with statistics, one block row', 1, 8, 70, 6, 8, 16, True),
    ('generator head data-gradient shape: 16 padded channels (3 real) -> 64 rows in four row groups', 2, 13, 70, 3, 64, 64, False),
    ('encoder head data-gradient shape: 5 real of 16 channels -> 16 rows', 1, 21, 45, 5, 16, 16, False),
])
def test_head_mfma_statistics_and_row_groups(case):
    """r06 (ABI 7): (a) the InstanceNorm statistics epilogue -- per (image, channel) sum and sum of squares of bias + conv, all
    SDN_STAT_SLOTS slots added, against float64 (1e-5 relative to the channel's own scale); (b) Cop = 64 over a 16-channel input:
    four row groups of 16 output channels from one staged patch.  Weights through the product's Stage.head_mfma."""
    import ctypes

    import torch.nn as nn
    from sdn_hip import check, lib, ptr, stream
    from sdn_hip import conv as hc
    from sdn_hip import convplan as cp
    name, N, H, W, cin, cout, cop, with_stats = case
    torch.manual_seed(900 + cin + cout)
    conv = nn.Conv2d(cin, cout, 7, padding=3).to(DEV)
    with torch.no_grad():
        conv.weight.mul_(2.0)
    x = torch.randn(N, cin, H, W)
    ref = F.conv2d(x.double(), conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=3)
    st = hc.Stage('conv', conv, 0)
    launches, (OH, OW) = cp.conv_fwd(7, 1, 3, H, W)
    Lh = launches[0]
    e = st.head_mfma('fwd', Lh.taps, Lh.tapidx, 16)
    e.refresh()
    KH, KW, dy_min, dx_min, R = e.meta
    assert R == cout and e.buf.numel() == (cop // 16) * ((49 * 2 + 3) // 4) * 2 * 64 * 8
    xg = _cl(x, 16)
    bias = torch.zeros(cop, device=DEV)
    bias[:cout] = conv.bias.detach()
    out = torch.full((N, OH, OW, cop), float('nan'), device=DEV)
    stats = torch.zeros(N, 8, cop, 2, dtype=torch.float64, device=DEV) if with_stats else None
    check(lib().sdn_conv_head_mfma(ptr(xg), N, H, W, 16, ptr(out), OH, OW, cop, cout, ptr(e.buf), KH, KW, dy_min, dx_min,
                                   0, 0, ptr(bias), 0, ptr(stats), stream()))
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).double().cpu()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err <= 5e-5, (name, err)
    if cout < cop:
        assert float(out[..., cout:].abs().max()) == 0.0
    if with_stats:
        s = stats.sum(1).cpu()                                   # [N, cop, 2]
        s1, s2 = ref.sum((2, 3)), (ref * ref).sum((2, 3))        # [N, cout]
        scale = (ref * ref).sum((2, 3)).sqrt() * (OH * OW) ** 0.5
        assert float(((s[:, :cout, 0] - s1).abs() / scale).max()) <= 1e-5, name
        assert float(((s[:, :cout, 1] - s2).abs() / s2).max()) <= 1e-5, name
        if cout < cop:
            assert float(s[:, cout:].abs().max()) == 0.0
