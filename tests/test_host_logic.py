"""Host-side logic that needs no GPU: synthetic meshes, camera constants, API surface."""
import numpy as np
import pytest
import torch

from oracle import nr_oracle as no
from sdn_hip import ops, synth


def test_perspective_width_matches_oracle_bitwise():
    for a in (30, 30.0, 14.833, np.arctan(384 / (2.0 * 725)) / np.pi * 180, 5.04):
        assert np.float32(ops.perspective_width(a)).tobytes() == np.float32(no.perspective_width(a)).tobytes()


def test_synth_meshes_are_closed_and_outward():
    for v, f in (synth.cube(), synth.uv_sphere(8, 12)):
        tri = v[f]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        c = tri.mean(1)
        assert (np.einsum('ij,ij->i', n, c) > 0).all()  # outward winding
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        und = np.sort(edges, 1)
        _, counts = np.unique(und, axis=0, return_counts=True)
        assert (counts == 2).all()  # closed 2-manifold


def test_car_like_budget_and_normalisation():
    v, f = synth.car_like(45000, seed=2)
    assert 38000 < len(f) < 52000 and 15000 < len(v) < 30000
    np.testing.assert_allclose(np.ptp(v, axis=0), 1.0, rtol=1e-5)
    assert f.min() == 0 and f.max() == len(v) - 1
    assert (f[:, 1] == f[:, 2]).sum() >= 24  # the deliberate degenerate faces


def test_neural_renderer_surface():
    import neural_renderer as nr
    for name in ['cross', 'get_points_from_angles', 'lighting', 'load_obj', 'look', 'look_at', 'Mesh', 'Adam',
                 'perspective', 'rasterize_rgbad', 'rasterize', 'rasterize_silhouettes', 'rasterize_depth',
                 'use_unsafe_rasterizer', 'Rasterize', 'Renderer', 'save_obj', 'vertices_to_faces']:
        assert hasattr(nr, name), name
    assert nr.__version__ == '1.1.3'
    r = nr.Renderer()
    assert (r.image_size, r.anti_aliasing, r.fill_back, r.camera_mode, r.near, r.far, r.rasterizer_eps) == \
        (256, True, True, 'look_at', 0.1, 100, 1e-3)
    np.testing.assert_allclose(r.eye, [0, 0, -(1. / np.tan(np.radians(30)) + 1)])
    assert nr.get_points_from_angles(2.0, 0, 0) == (0.0, 0.0, -2.0)


def test_cross_matches_numpy_and_oracle():
    import neural_renderer as nr
    a, b = torch.randn(50, 3), torch.randn(50, 3)
    np.testing.assert_allclose(nr.cross(a, b).numpy(), np.cross(a.numpy(), b.numpy()), rtol=1e-5, atol=1e-6)
    assert torch.equal(nr.cross(a, b), no.cross(a, b))


def test_obj_roundtrip(tmp_path):
    import neural_renderer as nr
    v, f = synth.uv_sphere(5, 6)
    p = str(tmp_path / 'm.obj')
    nr.save_obj(p, v, f)
    v2, f2 = nr.load_obj(p, normalization=False)
    np.testing.assert_allclose(v2, v, atol=1e-4)
    assert np.array_equal(f2, f)
    v3, _ = nr.load_obj(p)
    assert abs(np.abs(v3).max() - 1.0) < 1e-6  # unit cube normalisation (load_obj.py:132-136)


def test_masked_adam_skips_zero_gradients():
    import neural_renderer as nr
    p = torch.nn.Parameter(torch.ones(4))
    opt = nr.Adam([p], alpha=0.1)
    p.grad = torch.tensor([1.0, 0.0, -2.0, 0.0])
    opt.step()
    assert p[1] == 1 and p[3] == 1 and p[0] < 1 and p[2] > 1


def test_masked_adam_follows_chainer_adam_rule():
    """neural_renderer/optimizers.py:9-39 on chainer 4.1.0's AdamRule: t counts from 1, the step size is
    alpha * sqrt(1 - beta2^t) / (1 - beta1^t) times the parameter's own `lr`, elements with a zero gradient are skipped
    (their moments too)."""
    import math
    import neural_renderer as nr
    alpha, b1, b2, eps = 0.05, 0.9, 0.999, 1e-8
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64))
    p.lr = 0.5
    opt = nr.Adam([p], alpha=alpha, beta1=b1, beta2=b2, eps=eps)
    grads = [[0.3, 0.0, -1.0], [0.1, 2.0, 0.0], [-0.7, 0.5, 0.25]]
    x, m, v = [1.0, -2.0, 0.5], [0.0] * 3, [0.0] * 3
    for t, g in enumerate(grads, 1):
        p.grad = torch.tensor(g, dtype=torch.float64)
        opt.step()
        lr = alpha * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * 0.5
        for i in range(3):
            if g[i] != 0.0:
                m[i] += (1 - b1) * (g[i] - m[i])
                v[i] += (1 - b2) * (g[i] * g[i] - v[i])
                x[i] -= lr * m[i] / (math.sqrt(v[i]) + eps)
        np.testing.assert_allclose(p.detach().numpy(), x, rtol=1e-12, atol=0)


def test_pretrained_weights_are_never_silently_random(monkeypatch, tmp_path):
    """derenderer.py:25 / networks.py:470 ask for pretrained weights: without a weight file that is an error unless
    random initialisation was asked for explicitly; with a file the weights are loaded."""
    from derender3d.models import resnet
    monkeypatch.delenv('SDN_ALLOW_RANDOM_INIT', raising=False)
    monkeypatch.delenv('SDN_RESNET18_WEIGHTS', raising=False)
    with pytest.raises(RuntimeError):
        resnet.resnet18(pretrained=True)
    monkeypatch.setenv('SDN_ALLOW_RANDOM_INIT', '1')
    with pytest.warns(RuntimeWarning):
        net = resnet.resnet18(pretrained=True)
    sd = {k: torch.full_like(v, 0.25) if v.dtype.is_floating_point else v for k, v in net.state_dict().items()}
    f = tmp_path / 'r18.pth'
    torch.save(sd, str(f))
    monkeypatch.delenv('SDN_ALLOW_RANDOM_INIT')
    monkeypatch.setenv('SDN_RESNET18_WEIGHTS', str(f))
    net2 = resnet.resnet18(pretrained=True)
    assert float(net2.layer3[1].conv2.weight.mean()) == 0.25


def test_derenderer_state_dict_keys():
    from derender3d.models.derenderer import Derenderer
    keys = set(Derenderer().state_dict().keys())
    for k in ('net.conv1.weight', 'net.bn1.running_mean', 'net.layer1.0.conv1.weight', 'net.layer2.0.downsample.0.weight',
              'net.layer4.1.bn2.bias', 'net.fc.weight', 'fc1.weight', 'fc2.bias', '_fc3.weight'):
        assert k in keys, k
    assert Derenderer().state_dict()['_fc3.weight'].shape == (1552, 256)


def test_ffd_constraint_matrix_equals_constrain():
    import torch
    from derender3d.models.transforms import FFD, constrain_batched
    g = 4
    cons = [FFD.Constraint.symmetry(axis=FFD.Constraint.Axis.z),
            FFD.Constraint.homogeneity(axis=FFD.Constraint.Axis.y, index=[0, 1])]
    eye = torch.eye(3 * g ** 3).reshape(-1, 3, g, g, g)
    C = constrain_batched(eye, cons, g).reshape(3 * g ** 3, -1)
    x = torch.randn(7, 3 * g ** 3)
    want = constrain_batched(x.reshape(7, 3, g, g, g), cons, g).reshape(7, -1)
    assert float((x @ C - want).abs().max()) < 1e-6


def test_vgg19_loads_a_torchvision_format_checkpoint(monkeypatch, tmp_path):
    """VGGLoss asks for torchvision's vgg19(pretrained=True) (networks.py:470): SDN_VGG19_WEIGHTS names the file, its
    `features.N.*` keys are mapped onto the slices; without it construction raises unless random init was asked for."""
    import os
    import sys
    tex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '3d-sdn_amd', 'textural')
    if tex not in sys.path:
        sys.path.insert(0, tex)
    from models import networks as N
    vgg = N.Vgg19()
    want = {}
    for k, (a, b) in enumerate(N._VGG19_SLICES, 1):
        for i in range(a, b):
            for suffix in ('weight', 'bias'):
                key = 'slice%d.%d.%s' % (k, i, suffix)
                if key in vgg.state_dict():
                    want['features.%d.%s' % (i, suffix)] = torch.full_like(vgg.state_dict()[key], 0.5 + i)
    want['classifier.0.weight'] = torch.zeros(3, 3)          # ignored, as torchvision's file carries it
    f = tmp_path / 'vgg19.pth'
    torch.save(want, str(f))
    monkeypatch.setenv('SDN_VGG19_WEIGHTS', str(f))
    path_keys = {('features.%d.' % i): ('slice%d.%d.' % (k, i)) for k, (a, b) in enumerate(N._VGG19_SLICES, 1) for i in range(a, b)}
    N._load_vgg(vgg, path_keys)
    assert float(getattr(vgg.slice3, '10').weight.mean()) == 10.5 and float(getattr(vgg.slice1, '0').bias.mean()) == 0.5
    monkeypatch.delenv('SDN_VGG19_WEIGHTS')
    monkeypatch.delenv('SDN_ALLOW_RANDOM_INIT', raising=False)
    with pytest.raises(RuntimeError):
        N._load_vgg(N.Vgg19(), path_keys)


def test_packed_weight_tags_follow_the_owning_optimizer():
    """sdn_hip.conv caches re-packed weights under a tag; fused optimizers do not advance `_version`, so every optimizer
    step stamps the parameters IT owns (the generator's step must leave the discriminator's packed weights valid)."""
    from sdn_hip import conv as hc
    a = torch.nn.Parameter(torch.ones(3))
    b = torch.nn.Parameter(torch.ones(3))
    oa, ob = torch.optim.SGD([a], lr=0.1), torch.optim.SGD([b], lr=0.1)
    a.grad, b.grad = torch.ones(3), torch.ones(3)
    ta, tb = hc._tag(a), hc._tag(b)
    oa.step()
    assert hc._tag(a) != ta and hc._tag(b) == tb
    ta = hc._tag(a)
    ob.step()
    assert hc._tag(a) == ta and hc._tag(b) != tb
    ta, tb = hc._tag(a), hc._tag(b)
    hc.invalidate_weight_caches()      # writes through .data: everything
    assert hc._tag(a) != ta and hc._tag(b) != tb


def test_per_frame_constants_are_cached_by_value_and_identity():
    """camera.perspective_width keeps the device copy per list of angles (by VALUE); FFDBank._class_rows keeps the face
    gather for as long as the caller passes the same, unmodified class tensor (by IDENTITY and version)."""
    from neural_renderer import camera
    a = [10.0, 20.0, 30.0]
    w1 = camera.perspective_width(a, 3, torch.device('cpu'))
    w2 = camera.perspective_width(list(a), 3, torch.device('cpu'))
    assert w1 is w2
    assert np.array_equal(w1.numpy(), np.asarray([ops.perspective_width(x) for x in a], np.float32))
    w3 = camera.perspective_width([10.0, 20.0, 31.0], 3, torch.device('cpu'))
    assert w3 is not w1 and float(w3[2]) == float(ops.perspective_width(31.0))
    from derender3d.models.transforms import FFD, FFDBank
    ffds, faces = [], []
    for k in range(3):
        v, f = synth.uv_sphere(6 + k, 8)
        ffds.append(FFD(torch.tensor(v)))
        faces.append(torch.tensor(f))
    bank = FFDBank(ffds, faces)
    cls = torch.tensor([2, 0, 1, 2])
    c1, f1 = bank._class_rows(cls)
    c2, f2 = bank._class_rows(cls)
    assert c1 is c2 and f1 is f2 and c1.dtype == torch.int32
    assert torch.equal(f1, bank.faces[cls])
    cls[0] = 1                                   # in-place change: new version, new rows
    c3, f3 = bank._class_rows(cls)
    assert f3 is not f1 and torch.equal(f3, bank.faces[cls])
    other = cls.clone()                          # another tensor with equal contents: not trusted, recomputed
    assert bank._class_rows(other)[1] is not f3


def test_class_choice_is_cached_per_probability_tensor_and_the_row_gather_equals_advanced_indexing():
    """r05 (derender3d/models/__init__.py): the optimisation loop of scripts/main.py:433-456 hands the same detached class
    probabilities to render() every iteration -- arg-max, its log and the flat row index are kept per tensor (identity + version
    counter + storage address), never for a tensor that carries a graph; the chosen coefficients are then ONE index_select on the
    flattened [n * classes] rows, with the values and the gradient of the reference's coeffs[arange(n), classes] (:161-166)."""
    import types

    from derender3d.models import Derenderer3d
    me = types.SimpleNamespace(training=False, _force_no_sample=False, _banks={})
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn(6, 8, generator=g), dim=1)
    P1, P2 = {}, {}
    Derenderer3d._classes(me, {'_class_probs': probs}, P1)
    Derenderer3d._classes(me, {'_class_probs': probs}, P2)
    assert P2['classes'] is P1['classes'] and P2['_class_log_probs'] is P1['_class_log_probs'] and P2['class_rows'] is P1['class_rows']
    assert torch.equal(P1['classes'], probs.argmax(dim=1)) and torch.equal(P1['_class_log_probs'], torch.log(probs.max(dim=1)[0]))
    assert torch.equal(P1['class_rows'], torch.arange(6) * 8 + P1['classes'])
    probs[0] = torch.tensor([0.0] * 7 + [1.0])         # in-place change: new version, new choice
    P3 = {}
    Derenderer3d._classes(me, {'_class_probs': probs}, P3)
    assert P3['classes'] is not P1['classes'] and int(P3['classes'][0]) == 7
    P4 = {}
    Derenderer3d._classes(me, {'_class_probs': probs.clone()}, P4)     # equal contents, another tensor: recomputed
    assert P4['classes'] is not P3['classes'] and torch.equal(P4['classes'], P3['classes'])
    Derenderer3d.invalidate_class_cache(me)
    P5 = {}
    Derenderer3d._classes(me, {'_class_probs': probs}, P5)
    assert P5['classes'] is not P3['classes']
    live = probs.clone().requires_grad_(True)          # with a graph (encoder training): nothing is kept
    soft = torch.softmax(live, dim=1)
    Derenderer3d._classes(me, {'_class_probs': soft}, {})
    P6 = {}
    Derenderer3d._classes(me, {'_class_probs': soft}, P6)
    hit = me.__dict__.get('_argmax_hit')
    assert P6['_class_log_probs'].requires_grad and (hit is None or hit[0]() is not soft)
    # the row gather against advanced indexing, values and gradient
    coeffs = torch.randn(6, 8, 12, generator=g, requires_grad=True)
    ref = coeffs[torch.arange(6), P5['classes']]
    (ref * torch.arange(12.0)).sum().backward()
    want, coeffs.grad = coeffs.grad.clone(), None
    got = coeffs.reshape(6 * 8, -1).index_select(0, P5['class_rows'])
    (got * torch.arange(12.0)).sum().backward()
    assert torch.equal(got, ref) and torch.equal(coeffs.grad, want)


def test_unsafe_rasterizer_switch_selects_k1_coverage():
    """neural_renderer.use_unsafe_rasterizer / NEURAL_RENDERER_UNSAFE (rasterize.py:13-16, 1060-1062) are no longer ignored:
    they select SDN_K1_COVERAGE for the forward calls (the HIP side is tests/test_gpu_k1_coverage.py)."""
    import subprocess
    import sys

    import importlib

    import neural_renderer as nr
    import sdn_hip
    from sdn_hip import ops
    rz = importlib.import_module('neural_renderer.rasterize')   # (the package also exports a FUNCTION called rasterize)
    assert sdn_hip.K1_COVERAGE == 4096 and not ops.k1_coverage()
    nr.use_unsafe_rasterizer(True)
    try:
        assert ops.k1_coverage() and rz.USE_UNSAFE_IMPLEMENTATION
    finally:
        nr.use_unsafe_rasterizer(False)
    assert not ops.k1_coverage()
    code = ('import sys; sys.path[:0] = %r; import neural_renderer as nr; from sdn_hip import ops; '
            'import importlib; rz = importlib.import_module("neural_renderer.rasterize"); '
            'print(int(ops.k1_coverage()), int(rz.USE_UNSAFE_IMPLEMENTATION))' % [p for p in sys.path if '3d-sdn_amd' in p])
    import os
    for val, want in (('1', '1 1'), ('0', '0 0')):
        env = dict(os.environ, NEURAL_RENDERER_UNSAFE=val)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
        assert out.stdout.strip().endswith(want), (val, out.stdout, out.stderr[-500:])


def test_tiled_data_gradient_plan():
    """conv._tile_dgrad_plan (r04): which stride-1 data gradients go to sdn_conv_tile, and with which K split of the tail tiles.
    The residual layers of the generator at the benchmark size (batch 4, 24 x 78 maps, reflect pad 1 -> a 26 x 80 gradient grid =
    8 x 256 + 32 positions per image) take 8 slices; deterministic mode, an accumulating target and small grids do not split."""
    import types

    from sdn_hip import conv as hc
    from sdn_hip import convplan as cp

    st = types.SimpleNamespace(kind='conv', s=1, cout=1024)
    launches, (gh, gw) = cp.conv_dgrad(3, 1, 1, 24, 78, True)
    assert (gh, gw) == (26, 80) and len(launches) == 1
    assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 1024, 1024, 3, False, False) == 8
    # ordered sums wanted / a gradient already in the target: no atomics, and then two rounds of tiles are not worth it
    assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 1024, 1024, 3, True, False) is None
    assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 1024, 1024, 3, False, True) is None
    # a grid that fills whole rounds as it is (zero padding: 24 x 78 = 7.3 tiles per image, 256 tiles for 256 CUs): tile, no split
    l0, (h0, w0) = cp.conv_dgrad(3, 1, 1, 24, 78, False)
    assert (h0, w0) == (24, 78)
    assert hc._tile_dgrad_plan(st, l0, 4, h0, w0, 1024, 1024, 3, False, False) == 0
    # narrow layers, plain bf16, 16-channel K steps: the r03 kernel
    assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 1024, 128, 3, False, False) is None
    assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 1024, 1024, 1, False, False) is None
    assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 48, 1024, 3, False, False) is None
    # stride-2 layers come as four phase launches: not this path
    l2, _ = cp.conv_dgrad(3, 2, 1, 48, 156, False)
    assert len(l2) == 4 and hc._tile_dgrad_plan(st, l2, 4, 48, 156, 1024, 512, 3, False, False) is None
    # the switch
    import os
    old = os.environ.get('SDN_TILE_KERNELS')
    os.environ['SDN_TILE_KERNELS'] = 'wfh'
    try:
        assert hc._tile_dgrad_plan(st, launches, 4, gh, gw, 1024, 1024, 3, False, False) is None
    finally:
        if old is None:
            os.environ.pop('SDN_TILE_KERNELS')
        else:
            os.environ['SDN_TILE_KERNELS'] = old


def test_forward_kernel_selection_at_the_benchmark_shapes():
    """Which MFMA kernel the executor picks for the generator's layers at batch 4, 384 x 1248 (conv._halo_fwd_ok / _tile_fwd_ok;
    sdn_conv_halo_blocks is host arithmetic, no GPU needed): the 1024-channel residual layers take the patch kernel, the wide
    stride-2 layers the tile kernel, everything with fewer than 256 output channels or a 16-channel K step the r03 kernel."""
    import types

    from sdn_hip import conv as hc
    from sdn_hip import convplan as cp
    st = types.SimpleNamespace(kind='conv', s=1)
    res, (oh, ow) = cp.conv_fwd(3, 1, 1, 24, 78)
    assert hc._halo_fwd_ok(st, res, 4, oh, ow, 1024, 1024, 3)
    assert hc._tile_fwd_ok(st, res, 4, 1024, 1024, 3)
    down, (oh, ow) = cp.conv_fwd(3, 2, 1, 48, 156)              # 512 -> 1024, stride 2
    assert (oh, ow) == (24, 78) and not hc._halo_fwd_ok(st, down, 4, oh, ow, 512, 1024, 3)
    assert hc._tile_fwd_ok(st, down, 4, 512, 1024, 3)
    stem, (oh, ow) = cp.conv_fwd(7, 1, 3, 384, 1248)            # 48 -> 64: narrow N, K steps of 16 channels
    assert not hc._tile_fwd_ok(st, stem, 4, 48, 64, 3) and not hc._halo_fwd_ok(st, stem, 4, oh, ow, 48, 64, 3)
    d4, (oh, ow) = cp.conv_fwd(4, 1, 2, 49, 157)                # discriminator 256 -> 512, 4x4 stride 1
    assert hc._halo_fwd_ok(st, d4, 4, oh, ow, 256, 512, 3)
    assert not hc._tile_fwd_ok(st, res, 4, 1024, 1024, 1)       # plain bf16 never takes the planes path
