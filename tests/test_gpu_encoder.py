"""The derender3d encoder on the HIP kernels (sdn_hip/bnnet.py, csrc/conv_bn.hip + the MFMA conv kernels):
  * every op against torch fp64 on the CPU (the op's definition), forward and backward;
  * the whole Derenderer against tests/golden/encoder_golden.npz -- the REFERENCE's Derenderer class
    (derenderer.py:7-65) on seeded weights, eval and train mode, outputs / gradients / running statistics.
Tolerances: activations 1e-3 relative (north_star's gate for the fp32 networks; measured values are printed), gradients
relative L2 as stated per test."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'encoder_golden.npz'))
HEADS = ('_theta_deltas', '_translation2ds', '_log_scales', '_log_depths', '_class_probs', '_ffd_coeffs')


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize('cfg', [(3, 64, 7, 2, 3, 40, 52), (64, 64, 3, 1, 1, 17, 23), (64, 128, 3, 2, 1, 18, 26),
                                 (64, 128, 1, 2, 0, 18, 26), (256, 512, 1, 2, 0, 5, 7), (512, 512, 3, 1, 1, 2, 2)])
def test_conv2d_matches_float64(cfg):
    """bias-free ResNet convolutions (stem 7x7 s2, 3x3 s1 / s2, 1x1 s2 downsample): output, data and weight gradient."""
    from sdn_hip import bnnet as hb
    cin, cout, k, s, p, H, W = cfg
    torch.manual_seed(cin + cout + k)
    m = nn.Conv2d(cin, cout, k, s, p, bias=False)
    x = torch.randn(3, cin, H, W)
    xr = x.double().requires_grad_(True)
    wr = m.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    m = m.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = hb.conv2d(m, xg)
    assert y.shape == yr.shape
    y.backward(gy.float().to(DEV))
    assert rel(y, yr) <= 2e-5, rel(y, yr)
    assert rel(xg.grad, xr.grad) <= 2e-5, rel(xg.grad, xr.grad)
    assert rel(m.weight.grad, wr.grad) <= 2e-5, rel(m.weight.grad, wr.grad)


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('C,res,relu', [(64, False, True), (128, True, True), (512, True, False), (16, False, False)])
def test_batch_norm_matches_float64(C, res, relu, training):
    from sdn_hip import bnnet as hb
    torch.manual_seed(C + res + 2 * relu)
    bn = nn.BatchNorm2d(C)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.3)
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    N, H, W = 5, 9, 11
    x = torch.randn(N, C, H, W) * 1.7 + 0.3
    r = torch.randn(N, C, H, W) if res else None
    # float64 definition
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    g64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    rm, rv = bn.running_mean.double().clone(), bn.running_var.double().clone()
    yr = F.batch_norm(xr, rm, rv, g64, b64, training, 0.1, bn.eps)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    bn = bn.to(DEV)
    # channels-last storage viewed as NCHW, as the ops hand tensors to each other
    xg = x.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    rg = r.to(DEV).requires_grad_(True) if res else None
    y = hb.batch_norm(bn, xg, res=rg, relu=relu)
    y.backward(gy.float().to(DEV))
    assert rel(y, yr) <= 2e-6, rel(y, yr)
    assert rel(xg.grad, xr.grad) <= 2e-5, rel(xg.grad, xr.grad)
    assert rel(bn.weight.grad, g64.grad) <= 2e-5 and rel(bn.bias.grad, b64.grad) <= 2e-5
    if res:
        assert rel(rg.grad, rr.grad) <= 1e-6
    assert rel(bn.running_mean, rm) <= 2e-6 and rel(bn.running_var, rv) <= 2e-6
    assert int(bn.num_batches_tracked) == (1 if training else 0)


def test_max_pool_and_avg_pool_match_torch():
    from sdn_hip import bnnet as hb
    torch.manual_seed(5)
    for (N, C, H, W) in ((2, 64, 13, 16), (3, 16, 8, 7), (1, 128, 1, 5)):
        x = torch.randn(N, C, H, W)
        x[:, :, ::3, ::2] = 0.0   # ties (the ReLU in front of the pool produces runs of zeros)
        x = torch.relu(x)
        xr = x.double().requires_grad_(True)
        yr = F.max_pool2d(xr, 3, 2, 1)
        gy = torch.randn_like(yr)
        yr.backward(gy)
        xg = x.to(DEV).requires_grad_(True)
        y = hb.max_pool_3x3_s2(xg)
        y.backward(gy.float().to(DEV))
        assert torch.equal(y.detach().cpu(), yr.detach().float())
        # identical routing (ATen's first-maximum rule); an input pixel sums up to four window gradients, in another order
        assert rel(xg.grad, xr.grad) <= 1e-6 and bool(((xg.grad.cpu() != 0) == (xr.grad != 0)).all())
        ar = xr.detach().clone().requires_grad_(True)
        mr_ = ar.mean(dim=(2, 3))
        ga = torch.randn_like(mr_)
        mr_.backward(ga)
        ag = x.to(DEV).requires_grad_(True)
        a = hb.global_avg_pool(ag)
        a.backward(ga.float().to(DEV))
        assert rel(a, mr_) <= 1e-6 and rel(ag.grad, ar.grad) <= 1e-6


def _seeded():
    from derender3d.models.derenderer import Derenderer
    torch.manual_seed(int(GOLD['seed']))
    m = Derenderer()
    sd = m.state_dict()
    got = np.array([float(v.double().abs().sum()) for v in sd.values()])
    np.testing.assert_allclose(got, GOLD['checksums'], rtol=1e-6, atol=0)   # the golden file's weights
    return m


def _inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_encoder_golden', os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                     'golden', 'make_encoder_golden.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk, mk.inputs()


def test_backbone_stages_match_float64_oracle():
    """ResNet-18 feature stages on the HIP kernels vs the functional restatement in fp64 (eval and train mode): every
    BasicBlock output, 1e-4 relative."""
    from oracle import encoder_oracle as eo
    from sdn_hip import bnnet as hb
    mk, (images, mroi, droi, weights) = _inputs()
    for training in (False, True):
        m = _seeded()
        net = m.net
        net.train(training)
        sd64 = {k: v.double().clone() for k, v in net.state_dict().items()}
        taps = {}
        eo.resnet18_features(sd64, torch.tensor(images).double(), training=training, taps=taps)
        net = net.to(DEV)
        with torch.no_grad():
            x = hb.max_pool_3x3_s2(hb.batch_norm(net.bn1, hb.conv2d(net.conv1, torch.tensor(images, device=DEV)), relu=True))
            worst = rel(x, taps['stem'])
            assert worst <= 1e-4, ('stem', worst)
            for li, layer in enumerate((net.layer1, net.layer2, net.layer3, net.layer4), 1):
                for bi, block in enumerate(layer):
                    x = block(x)
                    e = rel(x, taps['layer%d.%d' % (li, bi)])
                    worst = max(worst, e)
                    assert e <= 1e-4, ('layer%d.%d' % (li, bi), training, e)
        print('backbone stages, training=%s: worst relative L2 %.2e' % (training, worst))


class _record_masks:
    """with _record_masks() as masks: ... -- the ReLU patterns (out > 0) of every fused batch_norm(..., relu=True) call, in
    call order, for evaluating the fp64 reference under the same activation pattern."""

    def __enter__(self):
        from sdn_hip import bnnet as hb
        self.hb, self.orig, self.masks = hb, hb.batch_norm, []

        def recording(bn, x, res=None, relu=False):
            y = self.orig(bn, x, res=res, relu=relu)
            if relu:
                self.masks.append((y.detach() > 0).cpu())
            return y
        hb.batch_norm = recording
        return self.masks

    def __exit__(self, *a):
        self.hb.batch_norm = self.orig
        return False


def test_stem_and_blocks_backward_match_float64():
    """Backward of the stem (conv7x7 s2 - bn - relu - maxpool) and of BasicBlocks with / without a downsample branch,
    train mode, against the same ops in fp64 under the same ReLU pattern (see _record_masks): input gradient and every
    parameter gradient, 1e-4 relative L2."""
    import copy
    from sdn_hip import bnnet as hb
    m = _seeded()
    net = m.net.train()
    torch.manual_seed(11)

    def block64(blk, x, masks):
        idt = x if blk.downsample is None else blk.downsample[1](blk.downsample[0](x))
        out = blk.bn1(blk.conv1(x)) * masks[0].double()
        return (blk.bn2(blk.conv2(out)) + idt) * masks[1].double()
    cases = [('stem', torch.randn(4, 3, 48, 56)), ('layer1.0', torch.relu(torch.randn(4, 64, 12, 14))),
             ('layer2.0', torch.relu(torch.randn(4, 64, 12, 14))), ('layer4.0', torch.relu(torch.randn(4, 256, 4, 6))),
             ('layer4.1', torch.relu(torch.randn(8, 512, 2, 2)))]
    for name, x in cases:
        dev = copy.deepcopy(net).to(DEV)
        dev.train()
        xg = x.to(DEV).requires_grad_(True)
        with _record_masks() as masks:
            if name == 'stem':
                y = hb.max_pool_3x3_s2(hb.batch_norm(dev.bn1, hb.conv2d(dev.conv1, xg), relu=True))
            else:
                li, bi = name.split('.')
                y = getattr(dev, li)[int(bi)](xg)
        ref = copy.deepcopy(net).double()
        ref.train()
        xr = x.double().requires_grad_(True)
        if name == 'stem':
            yr = F.max_pool2d(ref.bn1(ref.conv1(xr)) * masks[0].double(), 3, 2, 1)
            prefixes = ['conv1', 'bn1']
        else:
            yr = block64(getattr(ref, li)[int(bi)], xr, masks)
            prefixes = [name]
        gy = torch.randn_like(yr)
        yr.backward(gy)
        y.backward(gy.float().to(DEV))
        assert rel(y, yr) <= 1e-4, (name, rel(y, yr))
        assert rel(xg.grad, xr.grad) <= 1e-4, (name, 'input', rel(xg.grad, xr.grad))
        rp, dp = dict(ref.named_parameters()), dict(dev.named_parameters())
        for k in rp:
            if any(k.startswith(p + '.') for p in prefixes):
                assert dp[k].grad is not None, k
                e = rel(dp[k].grad, rp[k].grad)
                assert e <= 1e-4, (name, k, e)


def test_backbone_backward_under_identical_activation_patterns():
    """End-to-end backward of the ResNet-18 on the HIP kernels against the fp64 oracle evaluated with the SAME ReLU
    patterns (the masks the HIP forward produced): every parameter gradient and the input-side gradient, 1e-4 relative L2,
    train and eval mode.  This is the exact statement about the backward arithmetic (convolution data / weight
    gradients, BatchNorm, pooling); flips of the activation pattern are a forward-precision effect gated separately."""
    from oracle import encoder_oracle as eo
    from sdn_hip import bnnet as hb
    mk, (images, mroi, droi, weights) = _inputs()
    for training in (True, False):
        m = _seeded()
        net = m.net.train(training)
        sd64 = {k: (v.double().clone() if v.dtype.is_floating_point else v.clone()) for k, v in net.state_dict().items()}
        dev = net.to(DEV)
        with _record_masks() as recorded:
            x = torch.tensor(images, device=DEV)
            h = hb.max_pool_3x3_s2(hb.batch_norm(dev.bn1, hb.conv2d(dev.conv1, x), relu=True))
            for layer in (dev.layer1, dev.layer2, dev.layer3, dev.layer4):
                for block in layer:
                    h = block(h)
            pooled = hb.global_avg_pool(h)
        assert len(recorded) == 17
        w = torch.tensor(np.random.default_rng(3).normal(size=tuple(pooled.shape)))
        (pooled * w.float().to(DEV)).sum().backward()
        ps = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.dtype.is_floating_point and 'running' not in k
              and not k.startswith('fc.')}
        full = dict(sd64)
        full.update(ps)
        ref = eo.resnet18_features(full, torch.tensor(images).double(), training=training, relu_masks=recorded)
        assert rel(pooled, ref) <= 1e-4
        (ref * w).sum().backward()
        params = dict(dev.named_parameters())
        errs = sorted(((rel(params[k].grad, ps[k].grad), k) for k in ps), reverse=True)
        print('backbone backward (training=%s) under identical masks: worst %s %.1e, median %.1e'
              % (training, errs[0][1], errs[0][0], errs[len(errs) // 2][0]))
        assert errs[0][0] <= 1e-4, errs[:4]


def test_derenderer_matches_reference_golden_eval():
    mk, (images, mroi, droi, weights) = _inputs()
    m = _seeded().to(DEV).eval()
    t = lambda a: torch.tensor(a, device=DEV)
    with torch.no_grad():
        out = m(t(images), t(mroi), t(droi))
    worst = 0.0
    for k in HEADS:
        e = rel(out[k], torch.tensor(GOLD['eval' + k]))
        worst = max(worst, e)
        assert e <= 1e-3, (k, e)
    print('Derenderer eval vs the reference golden: worst relative L2 %.2e' % worst)


def test_derenderer_matches_reference_golden_train():
    """Train mode: outputs, the gradients the golden file holds (full small tensors, leading rows of large ones) and the
    running statistics after one forward.  Outputs: 1e-3 (measured 4e-5).  Gradients: the bf16x3 convolutions leave a
    forward difference of ~5e-5 after 17 layers, which flips about one ReLU per layer in these small tensors (layer4 is
    8 x 512 x 2 x 2 here); ONE flip there moves that layer's gradient by ~1e-2 relative L2 (measured with
    tests/gpu_net_diag.py: the gradient entering layer4.1 is exact, the one leaving it differs by 1.2e-2, while every
    block in isolation -- no flips -- agrees to 1e-5).  Hence a loose gate here (5e-2, cosine >= 0.999), and the tight
    statement about the backward kernels in test_backbone_backward_under_identical_activation_patterns."""
    mk, (images, mroi, droi, weights) = _inputs()
    m = _seeded().to(DEV).train()
    t = lambda a: torch.tensor(a, device=DEV)
    out = m(t(images), t(mroi), t(droi))
    worst = 0.0
    for k in HEADS:
        e = rel(out[k], torch.tensor(GOLD['train' + k]))
        worst = max(worst, e)
        assert e <= 1e-3, (k, e)
    sum((out[k] * t(weights[k])).sum() for k in HEADS).backward()
    params = dict(m.named_parameters())
    gworst, cworst = 0.0, 1.0
    for k in mk.GRAD_KEYS:
        a, b = mk.slice_of(k, params[k].grad).double().cpu().flatten(), torch.tensor(GOLD['grad/' + k]).double().flatten()
        gworst = max(gworst, rel(a, b))
        cworst = min(cworst, float((a * b).sum() / (a.norm() * b.norm())))
    print('Derenderer train vs the reference golden: outputs %.2e, gradients %.2e, cosine %.6f' % (worst, gworst, cworst))
    assert gworst <= 5e-2 and cworst >= 0.999, (gworst, cworst)
    sd = m.state_dict()
    for k in ('net.bn1.running_mean', 'net.bn1.running_var', 'net.layer4.1.bn2.running_mean', 'net.layer4.1.bn2.running_var',
              'net.layer2.0.downsample.1.running_var'):
        assert rel(sd[k], torch.tensor(GOLD['after/' + k])) <= 1e-4, k
    assert int(sd['net.bn1.num_batches_tracked']) == int(GOLD['after/net.bn1.num_batches_tracked'])


def test_derenderer_with_weights_loaded_from_a_torchvision_layout_checkpoint(monkeypatch, tmp_path):
    """VERDICT r05 missing #5: the PRETRAINED-weight route of the encoder (derenderer.py:25 `resnet18(pretrained=True)`) exercised on
    the device.  No torchvision and no network exist here, so the checkpoint is synthetic: a state_dict with torchvision 0.2.1's
    resnet18 keys and shapes (the restated module of oracle/encoder_oracle.py: conv1 ... layer4.1.bn2, the 1000-way `fc` the
    reference then replaces, `num_batches_tracked` counters), seeded values with non-trivial running statistics, written with
    torch.save.  SDN_RESNET18_WEIGHTS names the file, random initialisation is NOT allowed: the constructor must load it.  The
    device forward of the resulting Derenderer (eval mode, as scripts/main.py runs it) must then equal the float64 oracle evaluated
    on the FILE's tensors: 1e-3 relative per head (measured ~1e-5) -- a key that was skipped, transposed or left at its random
    initial value would be O(1)."""
    from derender3d.models.derenderer import Derenderer
    from oracle import encoder_oracle as eo
    torch.manual_seed(4242)
    ref = eo.RefResNet18()
    sd = {}
    for k, v in ref.state_dict().items():
        if k.endswith('running_var'):
            sd[k] = torch.rand_like(v) * 0.8 + 0.6
        elif k.endswith('running_mean'):
            sd[k] = torch.randn_like(v) * 0.1
        elif k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(1234, dtype=v.dtype)
        elif k.endswith('bn1.weight') or k.endswith('bn2.weight') or k.endswith('downsample.1.weight'):
            sd[k] = torch.rand_like(v) * 0.5 + 0.75
        else:
            sd[k] = v.clone() + 0.01 * torch.randn_like(v)
    assert sd['fc.weight'].shape == (1000, 512) and 'layer2.0.downsample.0.weight' in sd and len(sd) == 122
    path = tmp_path / 'resnet18-synthetic.pth'
    torch.save(sd, str(path))
    monkeypatch.delenv('SDN_ALLOW_RANDOM_INIT', raising=False)
    monkeypatch.setenv('SDN_RESNET18_WEIGHTS', str(path))
    torch.manual_seed(7)
    m = Derenderer()                                   # would raise without the file: random initialisation is not allowed
    got = m.state_dict()
    for k, v in sd.items():
        if k.startswith('fc.'):
            continue                                   # derenderer.py:27 replaces the classifier by Linear(512, 256)
        assert torch.equal(got['net.' + k], v), k
    full = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in got.items()}
    images = torch.randn(6, 3, 96, 128)
    mroi, droi = torch.rand(6, 2), torch.rand(6, 2) + 0.2
    want = eo.derenderer_forward(full, images.double(), mroi.double(), droi.double(), training=False)
    m = m.to(DEV).eval()
    with torch.no_grad():
        out = m(images.to(DEV), mroi.to(DEV), droi.to(DEV))
    worst = 0.0
    for k in HEADS:
        e = rel(out[k], want[k])
        worst = max(worst, e)
        assert e <= 1e-3, (k, e)
    print('Derenderer on a torchvision-layout checkpoint (SDN_RESNET18_WEIGHTS): worst head relative L2 %.2e' % worst)
