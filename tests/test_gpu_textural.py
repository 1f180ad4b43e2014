"""GPU parity of the textural conv stack (HIP kernels behind the C ABI) against the CPU oracle and the reference goldens.

Tolerances (written here, per BASELINE.json): activations within 1e-3 relative (we measure relative L2 per tensor AND
max-abs relative to the tensor's max); gradients of single layers and of the shallow golden networks within 1e-3 as
well, gradients through the full 28-stage generator within 5e-2 (ReLU mask flips, explained at that test).  The default
precision (bf16x3) is typically at 4e-6 per layer; the plain-bf16 mode is only checked to run and stay within 5e-2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural'), os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, 'tests', 'golden', 'textural_golden.npz')
REL = 1e-3


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rel_max(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def close(a, b, tol=REL, what=''):
    assert tuple(a.shape) == tuple(b.shape), (what, a.shape, b.shape)
    e2, em = rel_l2(a, b), rel_max(a, b)
    assert e2 <= tol and em <= tol, '%s: rel L2 %.3e, rel max %.3e (tol %.1e)' % (what, e2, em, tol)


# ---------------------------------------------------------------------------------------------------- single layers
# (name, module factory, cin, H, W): every conv flavour of networks.py with channel counts that hit all three N tiles
# (32 / 64 / 128+), partial M tiles, K that is not a multiple of 32, and both padding modes
def _layers():
    return [
        ('c7_reflect_3to20', lambda: [nn.ReflectionPad2d(3), nn.Conv2d(3, 20, 7)], 3, 13, 17),
        ('c7_reflect_48to64', lambda: [nn.ReflectionPad2d(3), nn.Conv2d(48, 64, 7)], 48, 12, 20),
        ('c7_reflect_64to3_tanh', lambda: [nn.ReflectionPad2d(3), nn.Conv2d(64, 3, 7), nn.Tanh()], 64, 10, 14),
        ('c3_s2_zero_40to136', lambda: [nn.Conv2d(40, 136, 3, 2, 1)], 40, 13, 18),
        ('c3_s2_zero_even', lambda: [nn.Conv2d(16, 32, 3, 2, 1)], 16, 16, 24),
        ('c3_reflect_128to128', lambda: [nn.ReflectionPad2d(1), nn.Conv2d(128, 128, 3)], 128, 12, 39),
        ('c3_zero_s1', lambda: [nn.Conv2d(24, 48, 3, 1, 1)], 24, 9, 11),
        ('c4_s2_p2_18to64_lrelu', lambda: [nn.Conv2d(18, 64, 4, 2, 2), nn.LeakyReLU(0.2, True)], 18, 21, 30),
        ('c4_s1_p2_64to1', lambda: [nn.Conv2d(64, 1, 4, 1, 2)], 64, 9, 12),
        ('c4_s1_p2_32to72', lambda: [nn.Conv2d(32, 72, 4, 1, 2)], 32, 7, 10),
        ('convT_64to32', lambda: [nn.ConvTranspose2d(64, 32, 3, 2, 1, 1)], 64, 6, 9),
        ('convT_144to136', lambda: [nn.ConvTranspose2d(144, 136, 3, 2, 1, 1)], 144, 5, 7),
        ('c3_s2_in_relu', lambda: [nn.Conv2d(16, 32, 3, 2, 1), nn.InstanceNorm2d(32), nn.ReLU(True)], 16, 12, 18),
        ('c4_s2_in_lrelu', lambda: [nn.Conv2d(16, 32, 4, 2, 2), nn.InstanceNorm2d(32), nn.LeakyReLU(0.2, True)], 16, 12, 18),
        ('convT_in_relu', lambda: [nn.ConvTranspose2d(32, 16, 3, 2, 1, 1), nn.InstanceNorm2d(16), nn.ReLU(True)], 32, 6, 8),
    ]


def _reference(mods, x):
    """the same torch modules, evaluated by torch on the CPU in float64: the per-layer oracle"""
    h = x
    for m in mods:
        h = m(h)
    return h


@pytest.mark.parametrize('case', _layers(), ids=lambda c: c[0])
def test_single_layer_forward_backward(case):
    from sdn_hip import conv as hc
    name, make, cin, H, W = case
    torch.manual_seed(11)
    mods = make()
    for m in mods:
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.normal_(m.weight, 0, 0.1)
            nn.init.normal_(m.bias, 0, 0.1)
    x = torch.randn(2, cin, H, W)
    # ---- oracle: float64 CPU
    import copy
    mods64 = [copy.deepcopy(m).double() for m in mods]
    x64 = x.double().requires_grad_(True)
    y64 = _reference(mods64, x64)
    w = torch.randn(y64.shape, dtype=torch.float64)
    (y64 * w).sum().backward()
    # ---- HIP
    gm = [copy.deepcopy(m).cuda() for m in mods]
    stages, last = hc.compile_sequential(gm)
    chain = hc.ConvChain(stages, [last], cin)
    xg = x.cuda().requires_grad_(True)
    yg = chain(xg)[0]
    close(yg, y64, what=name + ' forward')
    (yg * w.float().cuda()).sum().backward()
    close(xg.grad, x64.grad, what=name + ' grad input')
    conv_g = [m for m in gm if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))][0]
    conv_r = [m for m in mods64 if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))][0]
    close(conv_g.weight.grad, conv_r.weight.grad, what=name + ' grad weight')
    has_norm = any(isinstance(m, nn.InstanceNorm2d) for m in mods)
    if has_norm:
        # a bias in front of InstanceNorm has exactly zero gradient; torch's own value is round-off noise
        assert float(conv_g.bias.grad.abs().max()) == 0.0
        assert float(conv_r.bias.grad.abs().max()) < 1e-9 * max(1.0, float(conv_r.weight.grad.abs().max()))
    else:
        close(conv_g.bias.grad, conv_r.bias.grad, what=name + ' grad bias')


def test_resnet_block_chain():
    """ReflectionPad + conv + IN + ReLU + ReflectionPad + conv + IN + skip, twice (networks.py:244-283), 128 channels."""
    import copy
    from models import networks as N
    from oracle import textural_oracle as to
    torch.manual_seed(5)
    norm = N.get_norm_layer('instance')
    blocks = [N.ResnetBlock(128, 'reflect', norm), N.ResnetBlock(128, 'reflect', norm)]
    seq = nn.Sequential(*blocks)
    for m in seq.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, 0, 0.05)
    sd = {('model.' + k): v.double() for k, v in seq.state_dict().items()}
    x = torch.randn(2, 128, 9, 13)
    x64 = x.double().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.endswith('weight') or k.endswith('bias')}
    full = dict(sd)
    full.update(ps)
    y64 = to._resblock(full, 'model.1', to._resblock(full, 'model.0', x64))
    w = torch.randn(y64.shape, dtype=torch.float64)
    (y64 * w).sum().backward()
    from sdn_hip import conv as hc
    g = copy.deepcopy(seq).cuda()
    stages, last = hc.compile_sequential(list(g))
    chain = hc.ConvChain(stages, [last], 128)
    xg = x.cuda().requires_grad_(True)
    yg = chain(xg)[0]
    close(yg, y64, what='resblocks forward')
    (yg * w.float().cuda()).sum().backward()
    close(xg.grad, x64.grad, what='resblocks grad input')
    for k, p in g.named_parameters():
        if k.endswith('weight'):
            close(p.grad, ps['model.' + k].grad, what='resblocks grad ' + k)


# ---------------------------------------------------------------------------------------------------- reference goldens
def _gold(prefix, path=None):
    z = np.load(path or GOLD)
    pick = lambda kind: {k.split('/', 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith('%s/%s/' % (prefix, kind))}
    return pick('sd'), pick('in'), pick('out'), pick('grad'), pick('gin')


def _load_fresh_stats(net, sd):
    """load the reference's weights; running statistics back to their initial values so that one forward reproduces the
    values the reference stored after ITS one forward"""
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, nn.InstanceNorm2d):
            m.reset_running_stats()


def _check_running(net, sd):
    for k, v in net.state_dict().items():
        if 'running_' in k:
            assert float((v.cpu() - sd[k]).abs().max()) <= 1e-4 * max(1.0, float(sd[k].abs().max())), k
        if 'num_batches' in k:
            assert int(v) == int(sd[k]), k


def _norm_fed_biases(net):
    """names of conv biases whose output goes straight into an InstanceNorm: their true gradient is exactly zero"""
    names = {id(p): k for k, p in net.named_parameters()}
    out = set()
    for m in net.modules():
        if isinstance(m, nn.Sequential):
            ch = list(m)
            for a, b in zip(ch, ch[1:]):
                if isinstance(a, (nn.Conv2d, nn.ConvTranspose2d)) and isinstance(b, nn.InstanceNorm2d) and a.bias is not None:
                    out.add(names[id(a.bias)])
    return out


def _check_param_grads(net, grads):
    zero_bias = _norm_fed_biases(net)
    wmax = {k[:-len('.weight')]: float(v.abs().max()) for k, v in grads.items() if k.endswith('.weight')}
    for k, p in net.named_parameters():
        ref = grads[k]
        if k in zero_bias:
            # InstanceNorm removes the per-channel mean, so this gradient is identically zero; we return exact zeros,
            # the reference's autograd returns float round-off (orders of magnitude below the weight gradient)
            assert float(p.grad.abs().max()) == 0.0, k
            assert float(ref.abs().max()) < 1e-3 * wmax[k[:-len('.bias')]], k
            continue
        close(p.grad, ref, what='grad ' + k)


def test_generator_against_reference_golden():
    from models import networks as N
    sd, inp, out, grads, gin = _gold('G')
    G = N.define_G(6, 3, 8, 'global', n_downsample_global=2, n_blocks_global=2).cuda()
    assert set(G.state_dict().keys()) == set(sd.keys())
    _load_fresh_stats(G, sd)
    x = inp['x'].cuda().requires_grad_(True)
    y = G(x)
    close(y, out['y'], what='G output')
    assert float((y.cpu() - out['y']).abs().max()) < 1e-4  # depth/normal-style absolute gate on a tanh output
    (y * inp['w'].cuda()).sum().backward()
    close(x.grad, gin['x'], what='G grad input')
    _check_param_grads(G, grads)
    _check_running(G, sd)


def test_local_enhancer_against_reference_golden():
    """LocalEnhancer (`define_G(..., 'local', ...)`, networks.py:156-206): the half-resolution global trunk, the enhancer
    level's down branch, the sum of the two and the up branch with the tanh head -- three fused chains and the pooled input
    pyramid -- against the REFERENCE module's output, input gradient and every parameter gradient."""
    import os
    from models import networks as N
    sd, inp, out, grads, gin = _gold('L', os.path.join(os.path.dirname(GOLD), 'textural_local_golden.npz'))
    L = N.define_G(6, 3, 4, 'local', n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2).cuda()
    assert list(L.state_dict().keys()) == list(sd.keys())
    _load_fresh_stats(L, sd)
    x = inp['x'].cuda().requires_grad_(True)
    y = L(x)
    close(y, out['y'], what='LocalEnhancer output')
    assert float((y.cpu() - out['y']).abs().max()) < 1e-4
    (y * inp['w'].cuda()).sum().backward()
    close(x.grad, gin['x'], what='LocalEnhancer grad input')
    _check_param_grads(L, grads)
    _check_running(L, sd)


def test_encoder_against_reference_golden():
    from models import networks as N
    sd, inp, out, grads, gin = _gold('E')
    E = N.define_G(3, 2, 4, 'encoder', n_downsample_global=2, isTrain=False).cuda()
    _load_fresh_stats(E, sd)
    x = inp['x'].cuda().requires_grad_(True)
    y = E(x, inp['inst'].clone().cuda())
    close(y, out['y'], what='E output')
    (y * inp['w'].cuda()).sum().backward()
    close(x.grad, gin['x'], what='E grad input')
    _check_param_grads(E, grads)
    # generate_feat_dict: one entry per (id * batch + n) with output_nc means, equal to the pooled map's values
    fd = E.generate_feat_dict(inp['x'].cuda(), inp['inst'].clone().cuda())
    assert sorted(fd.keys()) == [0, 1, 6, 14, 15] and all(len(v) == 2 for v in fd.values())


def test_discriminator_against_reference_golden():
    from models import networks as N
    sd, inp, out, grads, gin = _gold('D')
    D = N.define_D(5, 8, 3, 'instance', False, 2, True).cuda()
    assert set(D.state_dict().keys()) == set(sd.keys())
    _load_fresh_stats(D, sd)
    x = inp['x'].cuda().requires_grad_(True)
    res = D(x)
    loss = 0
    assert len(res) == 2 and all(len(s) == 5 for s in res)
    for s, scale in enumerate(res):
        for j, f in enumerate(scale):
            close(f, out['f%d_%d' % (s, j)], what='D feature %d/%d' % (s, j))
            loss = loss + (f * inp['w%d_%d' % (s, j)].cuda()).sum()
    loss.backward()
    close(x.grad, gin['x'], what='D grad input')
    _check_param_grads(D, grads)
    _check_running(D, sd)


def test_three_scale_discriminator_against_reference_golden():
    """`--num_D 3`, the discriminator of BASELINE configs[3]: all 15 feature maps, the input gradient, every parameter
    gradient and the running statistics against the REFERENCE's own 3-scale MultiscaleDiscriminator
    (tests/golden/make_textural_golden_d3.py); the two coarse columns run on their side streams."""
    from models import networks as N
    sd, inp, out, grads, gin = _gold('D3', os.path.join(os.path.dirname(GOLD), 'textural_d3_golden.npz'))
    D = N.define_D(7, 8, 3, 'instance', False, 3, True).cuda()
    assert list(D.state_dict().keys()) == list(sd.keys())
    _load_fresh_stats(D, sd)
    x = inp['x'].cuda().requires_grad_(True)
    res = D(x)
    loss = 0
    assert len(res) == 3 and all(len(s) == 5 for s in res)
    for s, scale in enumerate(res):
        for j, f in enumerate(scale):
            close(f, out['f%d_%d' % (s, j)], what='D3 feature %d/%d' % (s, j))
            loss = loss + (f * inp['w%d_%d' % (s, j)].cuda()).sum()
    loss.backward()
    close(x.grad, gin['x'], what='D3 grad input')
    _check_param_grads(D, grads)
    _check_running(D, sd)


@pytest.mark.parametrize('hw', [(32, 48), (37, 51), (1, 5), (2, 2)])
def test_pyramid_pooling_gradient_for_strided_views(hw):
    """nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False) (networks.py:392) on the HIP kernels of
    csrc/fast_pool.hip: forward and backward equal torch's CPU implementation for NCHW tensors AND for channels-last-strided
    views (the generator's output as the discriminator receives it) -- the case in which torch's own GPU backward is wrong on
    this ROCm build (recorded, not asserted: a fixed torch must not break the test)."""
    from models import networks as N
    H, W = hw
    torch.manual_seed(3)
    pool = N._Pyramid()
    ref = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)
    base = torch.randn(2, H, W, 16)
    xc = base[..., :3].permute(0, 3, 1, 2).contiguous().clone().requires_grad_(True)     # CPU yardstick
    yc = ref(xc)
    w = torch.randn(yc.shape)
    (yc * w).sum().backward()
    for layout in ('nchw', 'channels_last_view'):
        if layout == 'nchw':
            x = xc.detach().cuda().requires_grad_(True)
            xin = x
        else:
            x = base.cuda().requires_grad_(True)
            xin = x[..., :3].permute(0, 3, 1, 2)
        y = pool(xin)
        assert tuple(y.shape) == tuple(yc.shape)
        assert float((y.detach().cpu() - yc.detach()).abs().max()) <= 1e-6, layout
        (y * w.cuda()).sum().backward()
        g = x.grad if layout == 'nchw' else x.grad[..., :3].permute(0, 3, 1, 2)
        assert float((g.cpu() - xc.grad).abs().max()) <= 1e-6, layout
        if layout != 'nchw':
            assert float(x.grad[..., 3:].abs().max()) == 0.0
            # what torch's own op does with the same view on this build
            x2 = base.cuda().requires_grad_(True)
            (ref(x2[..., :3].permute(0, 3, 1, 2)) * w.cuda()).sum().backward()
            print('torch avg_pool2d backward on the channels-last view, %dx%d: max abs error %.3e'
                  % (H, W, float((x2.grad[..., :3].permute(0, 3, 1, 2).cpu() - xc.grad).abs().max())))


@pytest.mark.parametrize('shape', [(2, 64, 25, 40), (1, 7, 3, 5), (3, 16, 9, 11)])
def test_fused_l1_loss_matches_torch(shape):
    """networks.L1Loss (criterionFeat, pix2pixHD_model.py:86) on the fused kernels against torch.nn.L1Loss in float64:
    value within 1e-6 relative, gradients (both operands) exact up to the 1/n scale's rounding; operands are NCHW views
    of channels-last storage, as the discriminator hands them over, with exact ties (sgn(0) = 0) planted."""
    from models import networks as N
    torch.manual_seed(3)
    n, c, h, w = shape
    a = torch.randn(n, h, w, c).cuda().permute(0, 3, 1, 2).requires_grad_(True)
    b = torch.randn(n, h, w, c).cuda().permute(0, 3, 1, 2)
    with torch.no_grad():
        b[0, 0, 0, :2] = a[0, 0, 0, :2]
    b.requires_grad_(True)
    from sdn_hip import ops
    assert ops.l1_loss_supported(a, b)
    loss = N.L1Loss()(a, b) * 3.0
    loss.backward()
    a64, b64 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ref = F.l1_loss(a64, b64) * 3.0
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    assert a.grad.stride() == a.stride()
    assert rel_max(a.grad, a64.grad) <= 1e-6 and rel_max(b.grad, b64.grad) <= 1e-6
    assert float(a.grad[0, 0, 0, 0]) == 0.0 and float(b.grad[0, 0, 0, 1]) == 0.0
    # layouts the fused kernels do not take fall through to torch's op (same value)
    c_ = torch.randn(n, c, h, w).cuda()
    assert not ops.l1_loss_supported(a, c_) or a.stride() == c_.stride()
    assert abs(float(N.L1Loss()(a.detach(), c_)) - float(F.l1_loss(a.detach(), c_))) <= 1e-6


def test_discriminator_dual_view_equals_two_passes(monkeypatch):
    """MultiscaleDiscriminator.forward_dual: ONE pass over the pyramid returned as two autograd views must equal the
    reference's two passes (pix2pixHD_model.py:191-193 on fake.detach(), :210 on fake) -- features, the
    discriminator's weight gradients (first view), the image gradient (second view) -- and leave the InstanceNorm
    running statistics where fake / real / fake forwards would.  Same kernels on the same data with ordered split-K
    sums: the gate is 1e-6 relative L2 (the fp64 statistics atomics may still differ in the last bit between two passes)."""
    import copy
    from models import networks as N
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    torch.manual_seed(17)
    D1 = N.define_D(5, 8, 3, 'instance', False, 2, True).cuda()
    D2 = copy.deepcopy(D1)
    label = torch.randn(2, 2, 40, 56).cuda()
    real = torch.randn(2, 3, 40, 56).cuda()
    img1 = torch.randn(2, 3, 40, 56).cuda().requires_grad_(True)
    img2 = img1.detach().clone().requires_grad_(True)

    def losses(res, seed):
        g = torch.Generator(device='cuda').manual_seed(seed)
        return sum((f * torch.randn(f.shape, generator=g, device='cuda')).sum() for s in res for f in s)
    # the reference's order: fake (detached), real, fake (attached)
    a_pool = D1([label, img1.detach()])
    a_real = D1([label, real])
    a_fake = D1([label, img1], detach_weights=True)
    b_pool, b_fake, second = D2.forward_dual([label, img2])
    b_real = D2([label, real])
    second()
    for ra, rb in ((a_pool, b_pool), (a_fake, b_fake), (a_real, b_real)):
        for sa, sb in zip(ra, rb):
            for fa, fb in zip(sa, sb):
                assert rel_l2(fb, fa) <= 1e-6
    # generator loss first (through the attached view), then the discriminator loss -- train.py:88-95
    losses(a_fake, 1).backward()
    losses(b_fake, 1).backward()
    assert rel_l2(img2.grad, img1.grad) <= 1e-6 and float(img1.grad.abs().max()) > 0
    gimg = img1.grad.clone()
    assert all(p.grad is None for p in D1.parameters()) and all(p.grad is None for p in D2.parameters())
    (losses(a_pool, 2) + losses(a_real, 3)).backward()
    (losses(b_pool, 2) + losses(b_real, 3)).backward()
    for (k, pa), pb in zip(D1.named_parameters(), D2.parameters()):
        assert pa.grad is not None and rel_l2(pb.grad, pa.grad) <= 1e-6, k
    assert torch.equal(img1.grad, gimg)       # the detached views added nothing
    assert rel_l2(img2.grad, img1.grad) <= 1e-6
    for (k, va), vb in zip(D1.state_dict().items(), D2.state_dict().values()):
        if 'running_' in k:
            assert float((va - vb).abs().max()) <= 1e-6 * max(1.0, float(va.abs().max())), k
        elif 'num_batches' in k:
            assert int(va) == int(vb), k


def test_discriminator_side_streams_change_nothing(monkeypatch):
    """MultiscaleDiscriminator runs its coarse columns on side HIP streams (SDN_D_STREAMS, default on) concurrently with
    the full-resolution one.  Features, input gradient and weight gradients must equal the single-stream execution
    (ordered split-K sums: 1e-6 relative), over several repetitions so that a missing stream dependency cannot hide."""
    import copy
    from models import networks as N
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    torch.manual_seed(23)
    D0 = N.define_D(5, 16, 3, 'instance', False, 3, True).cuda()
    label = torch.randn(2, 2, 96, 160).cuda()
    img = torch.randn(2, 3, 96, 160).cuda()

    def run(mode):
        monkeypatch.setenv('SDN_D_STREAMS', mode)
        D = copy.deepcopy(D0)
        x = img.clone().requires_grad_(True)
        res = D([label, x])
        g = torch.Generator(device='cuda').manual_seed(5)
        loss = sum((f * torch.randn(f.shape, generator=g, device='cuda')).sum() for s in res for f in s)
        loss.backward()
        torch.cuda.synchronize()
        return [f.detach() for s in res for f in s], x.grad, [p.grad for p in D.parameters()]
    ref = run('0')
    for _ in range(4):
        got = run('1')
        for a, b in zip(got[0], ref[0]):
            assert rel_l2(a, b) <= 1e-6
        assert rel_l2(got[1], ref[1]) <= 1e-6
        for a, b in zip(got[2], ref[2]):
            assert rel_l2(a, b) <= 1e-6


def test_weight_gradient_side_stream_changes_nothing(monkeypatch):
    """The weight-gradient launches of a backward pass go to a side HIP stream (SDN_WGRAD_STREAM, default on) beside the
    data-gradient chain.  Outputs and every gradient must equal the single-stream execution (ordered split-K sums: 1e-6
    relative), repeatedly, with an optimizer step in between so that buffers get recycled across the two streams."""
    import copy
    from models import networks as N
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    torch.manual_seed(29)
    G0 = N.define_G(6, 3, 16, 'global', n_downsample_global=2, n_blocks_global=3).cuda()
    x0 = torch.randn(2, 6, 64, 96).cuda()

    def run(mode, reps):
        monkeypatch.setenv('SDN_WGRAD_STREAM', mode)
        G = copy.deepcopy(G0)
        opt = torch.optim.SGD(G.parameters(), lr=1e-3)
        outs = []
        for r in range(reps):
            x = x0.clone().requires_grad_(True)
            y = G(x)
            g = torch.Generator(device='cuda').manual_seed(7 + r)
            opt.zero_grad()
            (y * torch.randn(y.shape, generator=g, device='cuda')).sum().backward()
            outs.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in G.parameters()]))
            opt.step()
        torch.cuda.synchronize()
        return outs
    ref, got = run('0', 4), run('1', 4)
    for (ya, xa, pa), (yb, xb, pb) in zip(got, ref):
        assert rel_l2(ya, yb) <= 1e-6 and rel_l2(xa, xb) <= 1e-6
        for a, b in zip(pa, pb):
            assert rel_l2(a, b) <= 1e-6


# ---------------------------------------------------------------------------------------------------- full architecture
def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def test_full_generator_activations_vs_oracle(monkeypatch):
    """The reference architecture (48 -> 3, ngf 64, 4 downsamplings, 9 blocks; Appendix C) at 64 x 96 against the fp64
    CPU oracle: EVERY stage's activation within 1e-3 relative (measured: 4e-6 after the stem, 6e-5 at the output).

    Gradients, two statements (the test runs with the ordered split-K sums, so every number below is reproducible):
      (1) backward arithmetic: against the fp64 oracle evaluated under the SAME activation pattern (the ReLU masks of the
          HIP forward): every weight gradient and the input gradient within 3e-4 relative L2 (measured 7.9e-5);
      (2) against the oracle's own pattern: a forward difference of ~6e-5 flips the sign of ~1e-5 of the pre-activations
          (about 14 of the 393k elements of the last 64-channel map), and every flipped unit changes that element's
          gradient by 100 %: measured 1.00e-2, gate 2e-2 and cosine >= 0.999.  The same happens between two fp64
          evaluations whose weights differ by 1e-5 relative (2e-2), i.e. it is the network's conditioning, not the
          kernels' arithmetic -- which (1) pins."""
    from models import networks as N
    from oracle import textural_oracle as to
    from sdn_hip import conv as hc
    # ordered split-K sums: the forward that is differentiated and the one that is inspected stage by stage below are then
    # the same numbers bit for bit (with atomics they differ by ~1e-5, i.e. by a few activation-pattern flips)
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    torch.manual_seed(2)
    G = N.define_G(48, 3, 64, 'global', 4, 9)
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    x = torch.randn(1, 48, 64, 96)
    acts = []
    xo = x.double().clone().requires_grad_(True)
    ps = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if k.endswith('weight') or k.endswith('bias')}
    full = dict(sd)
    full.update(ps)
    yo = to.global_generator(full, xo, 4, 9, collect=acts)
    w = torch.randn(yo.shape, dtype=torch.float64)
    (yo * w).sum().backward()
    G = G.cuda()
    xg = x.cuda().requires_grad_(True)
    yg = G(xg)
    chain = G._chain('model', G.model, 48)
    with torch.no_grad():
        ts, _ = chain.forward(xg.detach().permute(0, 2, 3, 1).contiguous(), hc.default_precision(), training=False)
    # oracle collects: stem, 4 down, 9 blocks, 4 up, head = 19 tensors; chain stages: 1 + 4 + 18 + 4 + 1
    stage_of = [1, 2, 3, 4, 5] + [5 + 2 * (b + 1) for b in range(9)] + [24, 25, 26, 27, 28]
    assert len(acts) == len(stage_of)
    worst = 0.0
    for a, si in zip(acts, stage_of):
        T = ts[si]
        t = T.data[..., :a.shape[1]].permute(0, 3, 1, 2)
        if T.relu:
            t = torch.relu(t)
        worst = max(worst, rel_l2(t, a))
        close(t, a, what='stage %d activation' % si)
    close(yg, yo, what='generator output')
    assert float((yg.detach().cpu().double() - yo.detach()).abs().max()) < 1e-4 * 10  # tanh output, absolute
    (yg * w.float().cuda()).sum().backward()
    # (1) the backward ARITHMETIC: fp64 oracle evaluated under the activation pattern of the HIP forward -- tight gate
    relu_stages = [1, 2, 3, 4, 5] + [6 + 2 * b for b in range(9)] + [24, 25, 26, 27]
    masks = []
    for si in relu_stages:
        T = ts[si]
        assert T.relu
        masks.append((T.data[..., :T.C] > 0).permute(0, 3, 1, 2).cpu())
    xm = x.double().clone().requires_grad_(True)
    pm = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if k.endswith('weight') or k.endswith('bias')}
    fullm = dict(sd)
    fullm.update(pm)
    ym = to.global_generator(fullm, xm, 4, 9, relu_masks=masks)
    (ym * w).sum().backward()
    worst_m = rel_l2(xg.grad, xm.grad)
    per_layer = ['x %.1e' % worst_m]
    for k, p in G.named_parameters():
        if k.endswith('weight'):
            e = rel_l2(p.grad, pm[k].grad)
            per_layer.append('%s %.1e' % (k.replace('model.', '').replace('.weight', '').replace('conv_block', 'cb'), e))
            worst_m = max(worst_m, e)
    print('same-pattern gradient error per layer:', ', '.join(per_layer))
    # (2) against the oracle's own activation pattern: dominated by the handful of flipped units (see the docstring)
    gtol, ctol = 2e-2, 0.999   # 2 x the (now reproducible) measured 1.00e-2
    worst_g = rel_l2(xg.grad, xo.grad)
    assert worst_g <= gtol and cosine(xg.grad, xo.grad) >= ctol, 'grad input rel L2 %.3e' % worst_g
    for k, p in G.named_parameters():
        if k.endswith('weight'):
            e = rel_l2(p.grad, ps[k].grad)
            worst_g = max(worst_g, e)
            assert e <= gtol and cosine(p.grad, ps[k].grad) >= ctol, 'grad %s rel L2 %.3e' % (k, e)
    print('worst stage activation rel L2 %.2e; worst gradient rel L2: same activation pattern %.2e, oracle pattern %.2e'
          % (worst, worst_m, worst_g))
    assert worst_m <= 3e-4, worst_m   # measured 7.9e-5


def test_forward_sees_the_weights_after_a_fused_optimizer_step():
    """The executor caches weights re-packed for the MFMA kernels.  torch's fused Adam (what Pix2PixHDModel builds) updates
    parameters WITHOUT advancing their `_version` counters, so the cache is also tagged with a process-wide optimizer-step
    epoch: after a step the forward must equal a freshly built network holding the same weights (r02 regression: the
    train step kept convolving with the first packed copy)."""
    from models import networks as N
    torch.manual_seed(31)
    G = N.define_G(6, 3, 8, 'global', n_downsample_global=2, n_blocks_global=2).cuda()
    x = torch.randn(2, 6, 32, 48).cuda()
    opt = torch.optim.Adam(G.parameters(), lr=5e-2, betas=(0.5, 0.999), fused=True)
    y0 = G(x)
    (y0 * torch.randn_like(y0)).sum().backward()
    versions = [p._version for p in G.parameters()]
    opt.step()
    assert [p._version for p in G.parameters()] == versions      # the premise: no version bump
    y1 = G(x).detach()
    fresh = N.define_G(6, 3, 8, 'global', n_downsample_global=2, n_blocks_global=2).cuda()
    fresh.load_state_dict(G.state_dict())
    y_ref = fresh(x).detach()
    assert rel_l2(y1, y_ref) <= 1e-5, rel_l2(y1, y_ref)
    assert rel_l2(y1, y0.detach()) > 1e-2                        # and the step did change the output
    # writes through .data: weights_init invalidates (reference idiom: net.apply(weights_init))
    G.apply(N.weights_init)
    fresh.load_state_dict(G.state_dict())
    assert rel_l2(G(x).detach(), fresh(x).detach()) <= 1e-5


def test_deterministic_mode_is_bit_reproducible(monkeypatch):
    """SDN_DETERMINISTIC=1 (or torch.use_deterministic_algorithms): the split-K partial sums of the data and weight
    gradient kernels are combined in slice order instead of with float atomics -- two runs of the same small generator
    step (whose layers all run split over K) give bit-identical outputs and gradients; and the ordered sums agree with the
    atomic ones to fp32 re-association."""
    from models import networks as N
    torch.manual_seed(4)
    G = N.define_G(12, 3, 32, 'global', 2, 3).cuda()
    x = torch.randn(1, 12, 32, 48).cuda()
    w = torch.randn(1, 3, 32, 48).cuda()

    def run():
        for p in G.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y = G(xi)
        (y * w).sum().backward()
        return [('y', y.detach().clone()), ('x.grad', xi.grad.clone())] + [(k, p.grad.clone()) for k, p in G.named_parameters()]
    atomic = run()
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    a, b = run(), run()
    diff = [(k, rel_l2(t0, t1)) for (k, t0), (_, t1) in zip(a, b) if not torch.equal(t0, t1)]
    assert not diff, diff
    for (k, t0), (_, t1) in zip(a, atomic):
        # vs the atomic sums (not reproducible themselves): a different rounding of the same forward -- ~1e-5 there; in
        # the gradients possibly a flipped unit or two, so only their direction is compared
        if k == 'y':
            assert rel_l2(t0, t1) <= 1e-4
        elif float(t1.abs().max()) > 0.0:
            assert cosine(t0, t1) >= 0.99, k


def test_full_size_properties():
    """BASELINE size (384 x 1248, the 375 x 1242 frame padded to a multiple of 16): properties that need no oracle.
    (a) batch independence: InstanceNorm networks treat images independently, G(cat[a, b]) == cat[G(a), G(b)];
    (b) every InstanceNorm stage's stored activation has mean 0 / variance 1 per (n, c);
    (c) the 3-scale discriminator's feature shapes follow the 4x4 / pad 2 / stride arithmetic of networks.py:420-437."""
    from models import networks as N
    from sdn_hip import conv as hc
    torch.manual_seed(9)
    G = N.define_G(48, 3, 64, 'global', 4, 9).cuda()
    x = torch.randn(2, 48, 384, 1248, device='cuda')
    with torch.no_grad():
        y2 = G(x)
        y0 = G(x[:1])
        assert tuple(y2.shape) == (2, 3, 384, 1248)
        # batch 1 runs the deep layers split over K (another summation order): equal to rounding, not bit for bit
        assert float((y2[:1] - y0).abs().max()) < 2e-4
        chain = G._chain('model', G.model, 48)
        ts, _ = chain.forward(x[:1].permute(0, 2, 3, 1).contiguous(), hc.default_precision(), training=False)
        for si, st in enumerate(chain.stages):
            if st.norm is None:
                continue
            t = ts[si + 1]
            xh = (t.xhat if t.xhat is not None else t.data)[..., :st.cout]
            m = xh.mean(dim=(1, 2))
            v = xh.var(dim=(1, 2), unbiased=False)
            assert float(m.abs().max()) < 1e-3, 'stage %d mean %g' % (si, float(m.abs().max()))
            assert float((v - 1).abs().max()) < 1e-2, 'stage %d var' % si
        D = N.define_D(18, 64, 3, 'instance', False, 3, True).cuda()
        res = D(torch.randn(1, 18, 384, 1248, device='cuda'))
        assert [tuple(f.shape[1:]) for f in res[0]] == [(64, 193, 625), (128, 97, 313), (256, 49, 157), (512, 50, 158),
                                                       (1, 51, 159)]
        assert len(res) == 3 and tuple(res[2][0].shape[2:]) == (49, 157)


def test_plain_bf16_mode_runs(monkeypatch):
    from models import networks as N
    from oracle import textural_oracle as to
    monkeypatch.setenv('SDN_CONV_PRECISION', '1')
    torch.manual_seed(4)
    G = N.define_G(6, 3, 8, 'global', 2, 2)
    sd = G.state_dict()
    x = torch.randn(1, 6, 32, 48)
    yo = to.global_generator(sd, x, 2, 2)
    y = G.cuda()(x.cuda())
    assert rel_l2(y, yo) < 5e-2


def test_cpu_input_raises():
    from models import networks as N
    G = N.define_G(6, 3, 8, 'global', 2, 2)
    with pytest.raises(NotImplementedError):
        G(torch.randn(1, 6, 16, 16))


# ---------------------------------------------------------------------------------------------------- model wrapper
def _small_model():
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    opt = default_options(gpu_ids=[0], label_nc=4, ngf=8, n_downsample_global=2, n_blocks_global=2, ndf=8, num_D=2,
                          nef=4, n_downsample_E=2, feat_num=2, feat_pose='1', feat_pose_num_bins=3, feat_normal='1',
                          no_vgg_loss=True, batchSize=2)
    torch.manual_seed(21)
    m = Pix2PixHDModel()
    m.initialize(opt)
    g = torch.Generator().manual_seed(22)
    n, h, w = 2, 32, 48
    label = torch.randint(0, 4, (n, 1, h, w), generator=g).float()
    inst = torch.zeros(n, 1, h, w)
    inst[0, :, 4:20, 6:30] = 1000
    inst[1, :, 10:28, 20:44] = 2000
    inst[1, :, 2:8, 2:12] = 1000
    pose = (inst > 0).float() * torch.randint(1, 4, (n, 1, h, w), generator=g).float()
    image = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    normal = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    return m, opt, [t.cuda() for t in (label, inst, image, pose, normal)]


def test_pix2pixhd_losses_match_oracle():
    """Pix2PixHDModel.forward (pix2pixHD_model.py:176-246): the eight losses equal the oracle's, computed from the same
    weights with the reference's layer arithmetic and loss weights (lambda_feat 5, lambda_L1 10, LSGAN)."""
    from oracle import textural_oracle as to
    m, opt, (label, inst, image, pose, normal) = _small_model()
    sdG = {k: v.detach().cpu().double() for k, v in m.netG.state_dict().items()}
    sdD = {k: v.detach().cpu().double() for k, v in m.netD.state_dict().items()}
    sdE = {k: v.detach().cpu().double() for k, v in m.netE.state_dict().items()}
    losses, fake = m.forward(label, inst.clone(), image, None, pose, normal, infer=True)
    # ---- oracle, fp64 CPU
    lab, ins, img, pos, nor = [t.cpu().double() for t in (label, inst, image, pose, normal)]
    one_hot = torch.zeros(2, 4, 32, 48, dtype=torch.float64).scatter_(1, lab.long(), 1.0)
    edge = torch.zeros(2, 1, 32, 48, dtype=torch.bool)
    edge[:, :, :, 1:] |= ins[:, :, :, 1:] != ins[:, :, :, :-1]
    edge[:, :, :, :-1] |= ins[:, :, :, 1:] != ins[:, :, :, :-1]
    edge[:, :, 1:, :] |= ins[:, :, 1:, :] != ins[:, :, :-1, :]
    edge[:, :, :-1, :] |= ins[:, :, 1:, :] != ins[:, :, :-1, :]
    input_label = torch.cat([one_hot, edge.double()], 1)
    feat = to.encoder(sdE, img, ins, 2)
    pose_oh = torch.zeros(2, 4, 32, 48, dtype=torch.float64).scatter_(1, pos.long(), 1.0)
    fake_o = to.global_generator(sdG, torch.cat([input_label, feat, pose_oh, nor], 1), 2, 2)
    close(fake, fake_o, what='fake image')
    pf = to.multiscale_discriminator(sdD, torch.cat([input_label, fake_o], 1), 2)
    pr = to.multiscale_discriminator(sdD, torch.cat([input_label, img], 1), 2)
    mse = lambda t, v: ((t - v) ** 2).mean()
    D_fake = sum(mse(s[-1], 0.0) for s in pf)
    D_real = sum(mse(s[-1], 1.0) for s in pr)
    G_GAN = sum(mse(s[-1], 1.0) for s in pf)
    feat_w = (4.0 / 4) * (1.0 / 2) * 5.0
    G_feat = sum(feat_w * (a - b).abs().mean() for sf, sr in zip(pf, pr) for a, b in zip(sf[:-1], sr[:-1]))
    G_L1 = (fake_o - img).abs().mean() * 10.0
    want = {'G_GAN': G_GAN, 'G_GAN_Feat': G_feat, 'D_real': D_real, 'D_fake': D_fake, 'G_L1': G_L1}
    got = dict(zip(m.loss_names, losses))
    for k, v in want.items():
        assert abs(float(got[k]) - float(v)) <= 1e-3 * max(1e-3, abs(float(v))), (k, float(got[k]), float(v))


def test_train_step_updates_and_skips_dead_work():
    """train.py:69-95 as one call: both optimizers step, losses stay finite over a few iterations, and the generator
    loss leaves no gradient on the discriminator (the reference computes and then discards it)."""
    m, opt, (label, inst, image, pose, normal) = _small_model()
    before_G = [p.detach().clone() for p in m.netG.parameters()]
    before_D = [p.detach().clone() for p in m.netD.parameters()]
    # generator loss alone must not touch D's gradients
    losses, _ = m.forward(label, inst.clone(), image, None, pose, normal)
    d = dict(zip(m.loss_names, losses))
    m.optimizer_D.zero_grad()
    (d['G_GAN'] + d['G_GAN_Feat'] + d['G_L1']).backward()
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in m.netD.parameters())
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in m.netG.parameters())
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in m.netE.parameters())
    hist = []
    for _ in range(3):
        out = m.train_step(label, inst.clone(), image, None, pose, normal)
        hist.append({k: float(v) for k, v in out.items()})
        assert all(np.isfinite(v) for v in hist[-1].values()), hist[-1]
    assert any(float((a - b.detach()).abs().max()) > 0 for a, b in zip(before_G, m.netG.parameters()))
    assert any(float((a - b.detach()).abs().max()) > 0 for a, b in zip(before_D, m.netD.parameters()))
    assert hist[-1]['G_L1'] < hist[0]['G_L1'] * 1.5  # not diverging


# ---------------------------------------------------------------------------------------------------- VGG loss
def _vgg_reference(vgg64, x):
    """torchvision-layout VGG19 slices (networks.py:467-497) evaluated by torch on the CPU in float64"""
    outs, h = [], x
    for n in range(1, 6):
        for m in getattr(vgg64, 'slice%d' % n):
            h = F.max_pool2d(h, 2, 2) if isinstance(m, nn.MaxPool2d) else (F.relu(h) if isinstance(m, nn.ReLU) else m(h))
        outs.append(h)
    return outs


def test_vgg19_features_and_loss_gradient():
    """Vgg19 relu{1..5}_1 features and VGGLoss = sum_i w_i L1(vgg_i(x), vgg_i(y).detach()) (networks.py:137-149) with
    random-init weights (the pretrained file cannot be downloaded): features within 1e-3, gradient wrt x within 5e-2
    relative L2 (L1's sign(x - y) and the ReLU masks flip where values nearly coincide; measured 1e-2) and cosine >= 0.998."""
    import copy
    from models import networks as N
    torch.manual_seed(9)
    loss_mod = N.VGGLoss([0])
    vgg = loss_mod.vgg
    for m in vgg.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
            nn.init.normal_(m.bias, 0, 0.05)
    vgg64 = copy.deepcopy(vgg).cpu().double()
    x, y = torch.randn(2, 3, 48, 80), torch.randn(2, 3, 48, 80)
    x64 = x.double().requires_grad_(True)
    fx, fy = _vgg_reference(vgg64, x64), _vgg_reference(vgg64, y.double())
    ref = sum(w * (a - b.detach()).abs().mean() for w, a, b in zip(loss_mod.weights, fx, fy))
    ref.backward()
    xg = x.cuda().requires_grad_(True)
    feats = vgg(xg)
    for i, (a, b) in enumerate(zip(feats, fx)):
        close(a, b, what='vgg relu%d_1' % (i + 1))
    loss = loss_mod(xg, y.cuda())
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref))
    loss.backward()
    g, g64 = xg.grad.double().cpu(), x64.grad
    assert rel_l2(g, g64) < 5e-2
    assert float((g * g64).sum() / (g.norm() * g64.norm())) > 0.998
    assert all(p.grad is None for p in vgg.parameters())  # frozen, as in the reference (requires_grad=False)


def test_generator_gradients_on_a_kink_free_case_need_no_activation_pattern():
    """An UNCONDITIONAL gradient gate: the tight gradient gates elsewhere evaluate the fp64 oracle under the
    activation pattern of the HIP forward, because a forward difference of 1e-5 flips the few ReLU units that sit within 1e-5 of
    their kink and each flip changes a gradient element by 100 %.  Here the ORACLE ALONE picks the case: seeds are tried (CPU,
    fp64) until a generator / input pair has no ReLU input closer than 1e-4 to zero -- twenty times the forward error of this
    depth -- so no unit can flip, the gradient is a smooth function of the forward values around the evaluation point, and the
    HIP gradients must match the oracle's own-pattern gradients tightly (3e-4 relative L2 per parameter, as the pattern-matched
    gates).  The loss is linear in the output (no |.| kink).  Nothing the product computed enters the reference side."""
    from models import networks as N
    from oracle import textural_oracle as to
    tau, found = 1e-4, None
    for seed in range(200, 400):
        torch.manual_seed(seed)
        G = N.define_G(6, 3, 8, 'global', n_downsample_global=2, n_blocks_global=2)
        sd = {k: v.clone() for k, v in G.state_dict().items()}
        x = torch.randn(1, 6, 24, 32)
        pre = []
        with torch.no_grad():
            to.global_generator({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, x.double(), 2, 2, preacts=pre)
        margin = min(float(t.abs().min()) for t in pre)
        if margin > tau:
            found = (seed, G, sd, x, margin, sum(t.numel() for t in pre))
            break
    assert found is not None, 'no kink-free case among 200 seeds'
    seed, G, sd, x, margin, units = found
    print('kink-free case: seed %d, %d ReLU units, closest to its kink %.2e' % (seed, units, margin))
    xo = x.double().clone().requires_grad_(True)
    ps = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if k.endswith('weight') or k.endswith('bias')}
    full = dict(sd)
    full.update(ps)
    yo = to.global_generator(full, xo, 2, 2)
    w = torch.randn(yo.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (yo * w).sum().backward()
    G = G.cuda()
    hc_inv = __import__('sdn_hip.conv', fromlist=['conv']).invalidate_weight_caches
    hc_inv()
    xg = x.cuda().requires_grad_(True)
    yg = G(xg)
    close(yg, yo, what='generator output')
    (yg * w.float().cuda()).sum().backward()
    worst = rel_l2(xg.grad, xo.grad)
    assert worst <= 3e-4, ('input gradient', worst)
    for k, p in G.named_parameters():
        ref = ps[k].grad
        if ref is None or float(ref.norm()) == 0.0:
            continue
        if k.endswith('bias') and float(ref.norm()) < 1e-9 * float(w.abs().sum()):
            continue       # a bias in front of InstanceNorm: exact zero gradient on both sides
        r = rel_l2(p.grad, ref)
        worst = max(worst, r)
        assert r <= 3e-4, (k, r)
    print('worst relative L2 over all gradients: %.2e' % worst)


def test_vgg_loss_against_the_reference_classes_golden():
    """The product's VGGLoss / Vgg19 (torchvision-compatible keys, own conv kernels) against tests/golden/vgg_golden.npz: loss
    and input gradient of the REFERENCE's own VGGLoss / Vgg19 classes (networks.py:137-149, 467-497) run in fp64 on the same
    seeded VGG19 (tests/golden/make_vgg_golden.py).  Loss 1e-4 relative; gradient 5e-2 relative L2 and cosine >= 0.998 (L1's
    sign and the ReLU masks flip where fp32 and fp64 values nearly coincide; measured ~1e-2); features 1e-3."""
    import numpy as np
    from models import networks as N
    from oracle import textural_oracle as to
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vgg_golden.npz'))
    sd = to.vgg19_seeded_state(int(z['seed']))
    loss_mod = N.VGGLoss([0])
    state = {}
    for n_, (a, b) in enumerate(to.VGG19_SLICES, 1):
        for i in range(a, b):
            for leaf in ('weight', 'bias'):
                k = 'features.%d.%s' % (i, leaf)
                if k in sd:
                    state['slice%d.%d.%s' % (n_, i, leaf)] = sd[k]
    loss_mod.vgg.load_state_dict(state, strict=True)
    loss_mod.vgg.cuda()
    from sdn_hip import conv as hc
    hc.invalidate_weight_caches()
    xg = torch.from_numpy(z['x']).float().cuda().requires_grad_(True)
    feats = loss_mod.vgg(xg)
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(z['feat%d_shape' % i])
        s = torch.from_numpy(z['feat%d_sample' % i])
        d = (f.detach().double().cpu().reshape(-1)[::97] - s).abs().max()
        assert float(d) <= 1e-3 * float(s.abs().max()), ('relu%d_1' % (i + 1), float(d))
    loss = loss_mod(xg, torch.from_numpy(z['y']).float().cuda())
    assert abs(float(loss) - float(z['loss'])) <= 1e-4 * abs(float(z['loss']))
    loss.backward()
    g, g64 = xg.grad.double().cpu(), torch.from_numpy(z['grad_x'])
    assert rel_l2(g, g64) < 5e-2
    assert float((g * g64).sum() / (g.norm() * g64.norm())) > 0.998


def test_split_k_layers_match_float64():
    """Layers whose M x N grid is far below 256 tiles run split over K (partial sums added with atomics, bias /
    statistics / activation in a second pass): batch-1 1024-channel residual-block conv with InstanceNorm, the 7x7 stem
    and a tanh head, against the same torch modules in float64."""
    import copy
    from sdn_hip import conv as hc
    torch.manual_seed(13)
    cases = [([nn.ReflectionPad2d(1), nn.Conv2d(256, 256, 3), nn.InstanceNorm2d(256), nn.ReLU(True)], 256, 12, 39),
             ([nn.ReflectionPad2d(3), nn.Conv2d(48, 64, 7), nn.InstanceNorm2d(64), nn.ReLU(True)], 48, 24, 40),
             ([nn.ReflectionPad2d(3), nn.Conv2d(64, 3, 7), nn.Tanh()], 64, 20, 24),
             ([nn.Conv2d(128, 1, 4, 1, 2)], 128, 9, 11)]
    for mods, cin, H, W in cases:
        for m in mods:
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.05)
                nn.init.normal_(m.bias, 0, 0.1)
        x = torch.randn(1, cin, H, W)
        mods64 = [copy.deepcopy(m).double() for m in mods]
        x64 = x.double().requires_grad_(True)
        y64 = _reference(mods64, x64)
        w = torch.randn(y64.shape, dtype=torch.float64)
        (y64 * w).sum().backward()
        gm = [copy.deepcopy(m).cuda() for m in mods]
        stages, last = hc.compile_sequential(gm)
        xg = x.cuda().requires_grad_(True)
        yg = hc.ConvChain(stages, [last], cin)(xg)[0]
        close(yg, y64, what='split-K forward %d' % cin)
        (yg * w.float().cuda()).sum().backward()
        close(xg.grad, x64.grad, what='split-K grad input %d' % cin)


@pytest.mark.parametrize('K', [7, 5000])
def test_segment_mean_matches_index_add(K):
    """sdn_segment_mean (Encoder instance pooling, networks.py:310-325) against a float64 index_add reference: the LDS
    table path (K <= 4096) and the global-atomics path, forward and backward."""
    from sdn_hip import ops
    g = torch.Generator().manual_seed(K)
    N, C, H, W = 2, 3, 37, 91
    seg = torch.randint(0, K, (N, H, W), generator=g)
    seg[0, :10, :40] = 0  # one large coherent segment, as instances are
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(N, C, H, W, generator=g)
    x64 = x.double().requires_grad_(True)
    flat = x64.permute(1, 0, 2, 3).reshape(C, -1)
    idx = seg.reshape(-1)
    sums = torch.zeros(C, K, dtype=torch.float64).index_add(1, idx, flat)
    cnt = torch.bincount(idx, minlength=K).double().clamp(min=1)
    ref = (sums / cnt)[:, idx].reshape(C, N, H, W).permute(1, 0, 2, 3)
    (ref * w.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    out, means = ops.SegmentMeanFn.apply(xg, seg.to(torch.int32).cuda(), K)
    assert float((out.double().cpu() - ref.detach()).abs().max()) < 1e-5
    present = torch.bincount(idx, minlength=K) > 0
    assert float((means.double().cpu() - (sums / cnt).detach())[:, present].abs().max()) < 1e-5
    (out * w.cuda()).sum().backward()
    assert float((xg.grad.double().cpu() - x64.grad).abs().max()) < 1e-5


def test_input_parts_restrict_the_data_gradient(monkeypatch):
    """A list of input tensors (what the reference concatenates: pix2pixHD_model.py:155-166, 199-210) gives the same
    output as the concatenation, and the gradient of the parts that require one equals the matching channel slice of
    the full input gradient; parts that do not require a gradient get none.  Run with the ordered split-K sums: the two
    forward passes are then the same numbers bit for bit, hence the same activation pattern, and the gradients (exact-fp32
    narrow kernel for the restricted range vs bf16x3 MFMA for the full one) agree to 1e-3."""
    from models import networks as N
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    torch.manual_seed(17)
    G = N.define_G(11, 3, 8, 'global', 2, 2).cuda()
    D = N.define_D(9, 8, 3, 'instance', False, 2, True).cuda()
    a, b, c = torch.randn(2, 4, 24, 40).cuda(), torch.randn(2, 5, 24, 40).cuda(), torch.randn(2, 2, 24, 40).cuda()
    w = torch.randn(2, 3, 24, 40).cuda()
    full = torch.cat((a, b, c), 1).requires_grad_(True)
    y_full = G(full)
    (y_full * w).sum().backward()
    bp = b.clone().requires_grad_(True)
    y = G([a, bp, c])
    assert torch.equal(y, y_full)
    (y * w).sum().backward()
    close(bp.grad, full.grad[:, 4:9], 1e-3, 'parts gradient')
    # discriminator: label part without gradient, image part with; pooled pyramid handled part by part
    lab, img = torch.randn(2, 6, 40, 56).cuda(), torch.randn(2, 3, 40, 56).cuda()
    fullD = torch.cat((lab, img), 1).requires_grad_(True)
    rf = D(fullD)
    sum(f.mean() for sc in rf for f in sc).backward()
    ip = img.clone().requires_grad_(True)
    rp = D([lab, ip], detach_weights=True)
    for sa, sb in zip(rp, rf):
        for fa, fb in zip(sa, sb):
            assert torch.equal(fa, fb)
    sum(f.mean() for sc in rp for f in sc).backward()
    close(ip.grad, fullD.grad[:, 6:9], 1e-3, 'D parts gradient')
