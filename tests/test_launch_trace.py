"""Host logic of the textural executor, pinned without a GPU through its launch trace (tests/trace_stub.py)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if p not in sys.path:
        sys.path.insert(0, p)

import trace_stub  # noqa: E402


@pytest.fixture
def trace(monkeypatch):
    return trace_stub.install(monkeypatch)


def _ptr(v):
    return v if isinstance(v, int) or v is None else getattr(v, 'value', v)


def test_generator_chain_wiring(trace):
    """GlobalGenerator(6 -> 3, ngf 8, 2 down, 2 blocks): 1 + 2 + 4 + 2 + 1 = 10 convolutions.  Forward: the stride-1/2
    convs are one gemm launch each, the two transposed convs four phase launches each, the 3-channel head the narrow
    kernel; every stage reads the buffer its predecessor wrote.  Backward: one weight gradient + one unpack per
    convolution, and nothing for the input (it does not require a gradient)."""
    from models import networks as N
    torch.manual_seed(0)
    G = N.define_G(6, 3, 8, 'global', n_downsample_global=2, n_blocks_global=2)
    x = torch.randn(2, 6, 32, 48)
    y = G(x)
    assert tuple(y.shape) == (2, 3, 32, 48)
    fwd = trace.names()
    # r06: the 7 x 7 stem (6 -> 8 channels under InstanceNorm: a 16-channel padded output) runs on the MFMA head kernel with its
    # statistics epilogue, like the 3-channel head (r05); their fragment-order weights are built by torch ops, not by pack records
    assert fwd.count('sdn_conv_gemm') == 2 + 4 + 2 * 4
    assert fwd.count('sdn_conv_narrow_fwd') + fwd.count('sdn_conv_head_mfma') == 2
    stem = [a for n_, a in trace.calls if n_ == 'sdn_conv_head_mfma'][0]
    assert _ptr(stem[19]), 'the stem under InstanceNorm must hand the head kernel a statistics buffer'
    assert fwd.count('sdn_in_apply') == 9                      # every conv but the head is followed by InstanceNorm
    assert fwd.count('sdn_conv_pack_weights') == 2 + 4 + 2 * 4
    # wiring: the input pointer of each gemm is the output pointer of an earlier launch (or the chain input)
    produced = set()
    first = True
    for name, a in trace.calls:
        if name == 'sdn_conv_gemm':
            src, dst = _ptr(a[0]), _ptr(a[5])
            assert first or src in produced, 'a gemm reads a buffer nothing wrote'
            first = False
            produced.add(dst)
        elif name == 'sdn_in_apply':
            produced.add(_ptr(a[0]))
            if _ptr(a[4]):
                produced.add(_ptr(a[4]))                       # residual blocks write x + xhat to a second buffer
    fwd_planes = [_ptr(a[15]) for n_, a in trace.calls if n_ == 'sdn_in_apply' and _ptr(a[15])]
    fwd_planes += [_ptr(a[3]) for n_, a in trace.calls if n_ == 'sdn_split_planes']
    trace.clear()
    y.sum().backward()
    bwd = trace.names()
    # r06: the two 7 x 7 layers (6 -> 8 and 8 -> 3: <= 16 channels on the d(out) side) take the MFMA head weight-gradient kernel
    assert bwd.count('sdn_conv_wgrad_head_mfma') == 2 and bwd.count('sdn_conv_wgrad_narrow') == 0
    assert bwd.count('sdn_conv_wgrad') + bwd.count('sdn_conv_wgrad_tile') + bwd.count('sdn_conv_wgrad_head_mfma') == 10
    # r04: the weight gradients of the MFMA layers read bf16 operand planes -- every plane pointer a tiled launch is handed
    # was written earlier: by the forward pass (the layer input: sdn_in_apply / sdn_split_planes) or by this pass (d loss /
    # d output: sdn_in_bwd / sdn_act_bwd)
    assert bwd.count('sdn_conv_wgrad_tile') == 8 and bwd.count('sdn_conv_wgrad') == 0
    planes = set(fwd_planes)
    for name, a in trace.calls:
        if name == 'sdn_in_bwd' and _ptr(a[8]):
            planes.add(_ptr(a[8]))
        elif name == 'sdn_act_bwd' and _ptr(a[6]):
            planes.add(_ptr(a[6]))
        elif name == 'sdn_conv_wgrad_tile':
            assert _ptr(a[0]) in planes and _ptr(a[2]) in planes, 'a tiled weight gradient reads planes nothing wrote'
    assert bwd.count('sdn_conv_unpack_grad') == 10
    assert bwd.count('sdn_in_bwd') == 9
    assert all(p.grad is not None for p in G.parameters())
    # data gradients: head 1, transposed convs 1 each, residual convs 1 each, stride-2 convs 4 phase launches each -- and
    # none for the stem, whose input needs no gradient
    # (r06: the head's data gradient -- dz in 16 padded channels towards 8 -> 16 padded input channels -- on the MFMA head kernel)
    assert bwd.count('sdn_conv_gemm') + bwd.count('sdn_conv_narrow_fwd') + bwd.count('sdn_conv_head_mfma') == 1 + 2 + 4 + 2 * 4
    assert bwd.count('sdn_conv_head_mfma') == 1
    assert bwd.count('sdn_reflect_fold') == 1 + 4              # adjoint of the reflection pads that take a data gradient


def test_dual_discriminator_pass_launches_one_forward(trace):
    """forward_dual: ONE forward over the pyramid; the weights' view back-propagates weight gradients, the input's view
    data gradients only (incl. the first layer's, which the detached view never needs)."""
    from models import networks as N
    torch.manual_seed(1)
    D = N.define_D(5, 8, 3, 'instance', False, 2, True)
    label, img = torch.randn(1, 2, 40, 56), torch.randn(1, 3, 40, 56, requires_grad=True)
    ref = N.define_D(5, 8, 3, 'instance', False, 2, True)
    ref([label, img.detach()])
    one_pass = trace.count('sdn_conv_gemm') + trace.count('sdn_conv_narrow_fwd')
    trace.clear()
    res_w, res_x, second = D.forward_dual([label, img])
    assert trace.count('sdn_conv_gemm') + trace.count('sdn_conv_narrow_fwd') == one_pass
    second()
    trace.clear()
    sum(f.sum() for s in res_x for f in s).backward()          # the generator's loss: into the image only
    assert trace.count('sdn_conv_wgrad') + trace.count('sdn_conv_wgrad_tile') + trace.count('sdn_conv_wgrad_narrow') == 0
    assert img.grad is not None and all(p.grad is None for p in D.parameters())
    n_dgrad_x = trace.count('sdn_conv_gemm') + trace.count('sdn_conv_narrow_fwd')
    trace.clear()
    sum(f.sum() for s in res_w for f in s).backward()          # the discriminator's loss: into the weights only
    assert trace.count('sdn_conv_wgrad') + trace.count('sdn_conv_wgrad_tile') + trace.count('sdn_conv_wgrad_narrow') == 2 * 5
    assert all(p.grad is not None for p in D.parameters())
    # ... and without the first layers' data gradients (one launch per stride-2 phase: 4 per column)
    assert trace.count('sdn_conv_gemm') + trace.count('sdn_conv_narrow_fwd') < n_dgrad_x


def test_an_optimizer_step_repacks_only_its_own_weights(trace):
    """The packed-weight caches are tagged per parameter with the step of the optimizer that owns it: after the
    generator's step a discriminator forward packs nothing, a generator forward everything again."""
    from models import networks as N
    torch.manual_seed(2)
    G = N.define_G(6, 3, 8, 'global', n_downsample_global=1, n_blocks_global=1)
    D = N.define_D(5, 8, 2, 'instance', False, 1, True)
    xg, xd = torch.randn(1, 6, 16, 24), torch.randn(1, 5, 16, 24)
    og, od = torch.optim.SGD(G.parameters(), lr=0.1), torch.optim.SGD(D.parameters(), lr=0.1)
    with torch.no_grad():
        G(xg)
        D(xd)
    packs_g = None
    trace.clear()
    with torch.no_grad():
        G(xg)
        D(xd)
    assert trace.count('sdn_conv_pack_weights') == 0            # cached
    G(xg).sum().backward()
    og.step()                                                   # plain SGD also bumps the parameters' versions
    trace.clear()
    with torch.no_grad():
        D(xd)
    assert trace.count('sdn_conv_pack_weights') == 0            # the generator's step left these alone
    with torch.no_grad():
        G(xg)
    packs_g = trace.count('sdn_conv_pack_weights')
    assert packs_g > 0
    sum(f.sum() for s in D(xd) for f in s).backward()
    od.step()
    trace.clear()
    with torch.no_grad():
        G(xg)
        assert trace.count('sdn_conv_pack_weights') == 0
        D(xd)
    assert trace.count('sdn_conv_pack_weights') > 0


def test_frame_step_launches(trace):
    """One 16-object frame of the optimisation loop (bench.make_step, tiny meshes): FFD decode, PerspectiveTransform and ONE
    call for the three maps each way -- and under the loop's silhouette-only loss a backward call without gradients for the
    normal and depth maps (autograd would hand over zero tensors)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, '3d-sdn_amd', 'geometric'))
    import bench
    old = bench.N_TRIS, bench.RENDER_SIZE
    bench.N_TRIS, bench.RENDER_SIZE = 300, 32
    try:
        dev = torch.device('cpu')
        bank, sizes, cls, params, targets, ptf = bench.build_scene(dev, seed=3)
        step = bench.make_step(dev, bank, cls, params, targets, ptf, backward=True)
        step()
        trace.clear()
        out = step()
    finally:
        bench.N_TRIS, bench.RENDER_SIZE = old
    assert tuple(out.shape) == (16, 5, 32, 32)
    n = trace.names()
    for name, cnt in (('sdn_ffd_decode', 1), ('sdn_ffd_coefficients', 2),   # constraints . coefficients, and its transpose
                      ('sdn_perspective_transform', 1), ('sdn_render_maps_fwd', 1),
                      ('sdn_render_maps_bwd', 1), ('sdn_perspective_transform_bwd', 1), ('sdn_ffd_decode_bwd', 1),
                      # the three maps are ONE C call each way (csrc/raster_maps.hip issues project / gather / normals /
                      # rasterize itself): none of the per-stage entry points is called from Python any more
                      ('sdn_project_vertices', 0), ('sdn_gather_faces', 0), ('sdn_face_normals', 0), ('sdn_rasterize_fwd', 0),
                      ('sdn_rasterize_bwd', 0), ('sdn_gather_faces_bwd', 0), ('sdn_project_vertices_bwd', 0)):
        assert n.count(name) == cnt, (name, n.count(name))
    # under the loop's silhouette-only loss the backward call gets no gradient for the normal and depth maps (autograd would
    # hand over zero tensors): sdn_render_maps_bwd then skips the colour / depth pass and the face-normal branch
    bwd = trace.of('sdn_render_maps_bwd')[0]
    g_alpha, g_normal, g_depth = bwd[18], bwd[19], bwd[20]     # (bwd[17] = the forward call's background colour, ABI 6)
    assert getattr(g_alpha, 'value', g_alpha) and not getattr(g_normal, 'value', g_normal) and not getattr(g_depth, 'value', g_depth)
    assert all(p.grad is not None for p in params.values())


def test_running_statistics_replay_equals_sequential_updates():
    """conv.update_running applies what one training-mode forward does to InstanceNorm2d(track_running_stats=True); the
    dual discriminator pass uses it twice around the real image's pass: fake, real, fake -- the reference's order."""
    import torch.nn as nn
    from sdn_hip import conv as hc
    torch.manual_seed(4)
    nm = nn.InstanceNorm2d(6, affine=False, track_running_stats=True)
    ref = nn.InstanceNorm2d(6, affine=False, track_running_stats=True)
    fake, real = torch.randn(3, 6, 9, 11), torch.randn(3, 6, 9, 11) * 2 + 1
    for x in (fake, real, fake):
        ref(x)

    def batch_stats(x):   # what the kernel leaves in the collection buffers: batch mean of the instance statistics
        return x.mean(dim=(2, 3)).mean(0), x.var(dim=(2, 3), unbiased=True).mean(0)
    bm, bv = batch_stats(fake)
    hc.update_running([(nm, bm, bv)])
    rm, rv = batch_stats(real)
    hc.update_running([(nm, rm, rv)])
    hc.update_running([(nm, bm, bv)])
    assert torch.allclose(nm.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(nm.running_var, ref.running_var, rtol=1e-5, atol=1e-6)


def test_plan_caches_are_bounded(trace):
    """inference on images of many sizes must not pile up compiled plans (each owns a C program): a chain keeps
    the PLAN_CACHE most recently used forward plans, their backward plans go with them, and a backward plan found under the
    id of a freed forward plan is not reused."""
    from models import networks as N
    from sdn_hip import conv as hc
    torch.manual_seed(2)
    E = N.define_G(3, 2, 4, 'encoder', 2)
    for k in range(hc.PLAN_CACHE + 5):
        x = torch.randn(1, 3, 16 + 4 * k, 24, requires_grad=True)
        inst = torch.zeros(1, 1, 16 + 4 * k, 24)
        E(x, inst).sum().backward()
    found = [c for m in E.modules() for c in m.__dict__.get('_chains', {}).values() if isinstance(c, hc.ConvChain)]
    assert found, 'no ConvChain behind the encoder'
    for c in found:
        assert 0 < len(c._fwd_plans) <= hc.PLAN_CACHE
        assert len(c._bwd_plans) <= 4 * hc.PLAN_CACHE
        live = set(id(p) for p in c._fwd_plans.values())
        assert all(k[0] in live for k in c._bwd_plans), 'a backward plan outlived its forward plan in the cache'


def test_encoder_shaped_chain_routes_its_7x7_layers_to_the_head_kernel(trace):
    """r06: define_G(3, 5, 16, 'encoder', 2) -- the stem (3 -> 16 under InstanceNorm) and the head (16 -> 5) are 7 x 7 layers with
    <= 16 channels on one side: forward on sdn_conv_head_mfma (the stem with a statistics buffer, the head without), and in the
    backward pass the head's data gradient (16 rows towards all 16 input channels) on the same kernel; SDN_HEAD_WIDE=0 keeps the
    r05 routing (stem and that data gradient on sdn_conv_gemm)."""
    from models import networks as N
    torch.manual_seed(5)
    E = N.define_G(3, 5, 16, 'encoder', 2)
    x = torch.randn(1, 3, 32, 48)
    inst = torch.zeros(1, 1, 32, 48)
    inst[:, :, 8:20, 10:30] = 1000
    y = E(x, inst)
    heads = [a for n_, a in trace.calls if n_ == 'sdn_conv_head_mfma']
    assert len(heads) == 2
    assert _ptr(heads[0][19]) and heads[0][9] == 16 and heads[0][4] == 16        # stem: statistics, 16 rows over 16 padded channels
    assert not _ptr(heads[1][19]) and heads[1][9] == 5                          # head: 5 rows, tanh, no norm
    trace.clear()
    y.sum().backward()
    bwd_heads = [a for n_, a in trace.calls if n_ == 'sdn_conv_head_mfma']
    assert len(bwd_heads) == 1 and bwd_heads[0][9] == 16 and bwd_heads[0][8] == 16   # 16 rows into a 16-channel gradient tensor


def test_eager_repack_leaves_the_next_pass_nothing_to_pack(trace):
    """r06: conv.eager_repack refreshes the packed weights of the plans a module's chains ran last -- Pix2PixHDModel.train_step
    calls it on a side stream right behind optimizer_G.step().  After it, neither the next forward nor the next backward pass
    of that module packs anything; another module's packs are untouched."""
    from models import networks as N
    from sdn_hip import conv as hc
    torch.manual_seed(6)
    G = N.define_G(6, 3, 8, 'global', n_downsample_global=1, n_blocks_global=1)
    D = N.define_D(5, 8, 2, 'instance', False, 1, True)
    xg, xd = torch.randn(1, 6, 16, 24, requires_grad=True), torch.randn(1, 5, 16, 24)
    og = torch.optim.SGD(G.parameters(), lr=0.1)
    G(xg).sum().backward()
    with torch.no_grad():
        D(xd)
    og.step()
    trace.clear()
    hc.eager_repack([G])
    packs = trace.count('sdn_conv_pack_weights') + trace.count('sdn_conv_pack_weights_kmajor')
    assert packs > 0
    trace.clear()
    G(xg).sum().backward()
    with torch.no_grad():
        D(xd)
    assert trace.count('sdn_conv_pack_weights') + trace.count('sdn_conv_pack_weights_kmajor') == 0
    trace.clear()
    hc.eager_repack([G, D])                                   # nothing is stale: nothing is launched
    assert len(trace.calls) == 0
