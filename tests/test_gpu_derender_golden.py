"""The derender3d decode on the GPU against values produced by the REFERENCE's own code.

tests/golden/derender_golden.npz holds what /root/reference/geometric/derender3d/models/transforms.py (FFD :68-99,
PerspectiveTransform :102-158) and models/__init__.py (Derenderer3d.render :94-250) computed for fixed inputs
(generator: tests/golden/make_derender_golden.py).  Here the HIP ops -- sdn_ffd_decode, the fused
sdn_perspective_transform, and the batched render path that chains them -- run on cuda:0 and are compared with those
values; their gradients are compared with the element-wise CPU path, i.e. the reference's arithmetic under autograd
(the forward of that path is pinned to the same golden file by tests/test_oracle_golden.py).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'derender_golden.npz'))


def _constraints():
    from derender3d.models.transforms import FFD
    return [FFD.Constraint.symmetry(axis=FFD.Constraint.Axis.z),
            FFD.Constraint.homogeneity(axis=FFD.Constraint.Axis.y, index=[0, 1])]


def _ffds():
    from derender3d.models.transforms import FFD
    return [FFD(torch.tensor(GOLD['template%d_vertices' % k]), constraints=_constraints()) for k in range(2)]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def test_ffd_bank_decode_matches_reference_golden():
    """sdn_ffd_decode (csrc/fast_ffd.hip) through FFDBank.decode: vertices of both templates within 1e-6 relative
    (abs floor 2e-6: vertex coordinates are O(0.5)) of the reference's FFD.forward; coefficient gradient vs the CPU
    element-wise FFD under autograd."""
    from derender3d.models.transforms import FFDBank
    ffds = _ffds()
    faces = [torch.tensor(GOLD['template%d_faces' % k]) for k in range(2)]
    bank = FFDBank(ffds, faces).to(DEV)
    # objects: template 0, template 1, template 1, template 0 (repeats exercise the class gather)
    order = [0, 1, 1, 0]
    coeff = torch.stack([torch.tensor(GOLD['ffd%d_coeff' % k]) for k in order])
    cg = coeff.to(DEV).requires_grad_(True)
    verts, f = bank.decode(cg, torch.tensor(order, device=DEV))
    assert verts.is_cuda and f.is_cuda
    w = []
    for i, k in enumerate(order):
        nv = GOLD['template%d_vertices' % k].shape[0]
        ref = GOLD['ffd%d_vertices' % k]
        got = verts[i, :nv].detach().cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=2e-6)
        # padded vertices repeat vertex 0 of the template (FFDBank docstring)
        if verts.shape[1] > nv:
            np.testing.assert_allclose(verts[i, nv:].detach().cpu().numpy(), np.broadcast_to(got[0], (verts.shape[1] - nv, 3)),
                                       rtol=0, atol=1e-7)
        nf = GOLD['template%d_faces' % k].shape[0]
        assert torch.equal(f[i, :nf].cpu(), faces[k])
        w.append(torch.linspace(-1, 1, nv * 3).reshape(nv, 3) * (i + 1))
    loss = sum((verts[i, :w[i].shape[0]] * w[i].to(DEV)).sum() for i in range(len(order)))
    loss.backward()
    for i, k in enumerate(order):
        c = coeff[i].clone().requires_grad_(True)
        (ffds[k](c) * w[i]).sum().backward()
        assert _rel(cg.grad[i].cpu(), c.grad) <= 1e-5, (i, _rel(cg.grad[i].cpu(), c.grad))


def _pt_args(device):
    t = lambda k: torch.tensor(GOLD[k], device=device)
    return dict(vertices=t('pt_vertices'), scales=t('pt_scales'), rotations=t('pt_rotations'),
                translations=t('pt_translations'))


def test_fused_perspective_transform_matches_reference_golden():
    """Test-time form (zoom-to-fit, transforms.py:147-158) = the fused HIP op: vertices and zooms vs the reference's
    outputs (1e-6 relative; z is O(100) after the zoom so the absolute floor is 1e-6 of the largest value), all five
    gradients vs the element-wise CPU path."""
    from derender3d.models.transforms import PerspectiveTransform
    pt = PerspectiveTransform()
    a = {k: v.requires_grad_(True) for k, v in _pt_args(DEV).items()}
    zt = torch.tensor(GOLD['pt_zoom_tos'], device=DEV).requires_grad_(True)
    out, zooms = pt(a['vertices'], scales=a['scales'], rotations=a['rotations'], translations=a['translations'],
                    perspective_translations=a['translations'], zoom_tos=zt)
    assert out.is_cuda and out.grad_fn is not None and 'PerspectiveTransformFn' in type(out.grad_fn).__name__
    ref = GOLD['pt_test_out']
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6 * float(np.abs(ref).max()))
    np.testing.assert_allclose(zooms.detach().cpu().numpy(), GOLD['pt_test_zooms'], rtol=1e-6, atol=0)
    rng = np.random.default_rng(5)
    w = torch.tensor(rng.normal(size=ref.shape).astype(np.float32))
    wz = torch.tensor(rng.normal(size=(2, 1)).astype(np.float32))
    ((out * w.to(DEV)).sum() + (zooms * wz.to(DEV)).sum()).backward()
    c = {k: v.requires_grad_(True) for k, v in _pt_args('cpu').items()}
    zc = torch.tensor(GOLD['pt_zoom_tos']).requires_grad_(True)
    oc, zoc = pt._forward_elementwise(c['vertices'], scales=c['scales'], rotations=c['rotations'],
                                      translations=c['translations'], perspective_translations=c['translations'],
                                      zoom_tos=zc)
    ((oc * w).sum() + (zoc * wz).sum()).backward()
    for k in a:
        assert _rel(a[k].grad.cpu(), c[k].grad) <= 2e-5, (k, _rel(a[k].grad.cpu(), c[k].grad))
    assert _rel(zt.grad.cpu(), zc.grad) <= 2e-5


def test_train_form_perspective_transform_matches_reference_golden():
    """Training form (given zoom, crop-centred shear, transforms.py:139-150; also what the optimisation loop of
    scripts/main.py:433-456 runs) = the fused HIP op with `zooms`: output vs the reference's, all six gradients vs the
    element-wise CPU path."""
    from derender3d.models.transforms import PerspectiveTransform
    pt = PerspectiveTransform()
    a = {k: v.requires_grad_(True) for k, v in _pt_args(DEV).items()}
    pg = torch.tensor(GOLD['pt_ptranslations'], device=DEV).requires_grad_(True)
    zg = torch.tensor(GOLD['pt_zooms'], device=DEV).requires_grad_(True)
    out = pt(a['vertices'], scales=a['scales'], rotations=a['rotations'], translations=a['translations'],
             perspective_translations=pg, zooms=zg)
    assert out.is_cuda and 'PerspectiveTransformFn' in type(out.grad_fn).__name__
    ref = GOLD['pt_train_out']
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6 * float(np.abs(ref).max()))
    w = torch.tensor(np.random.default_rng(6).normal(size=ref.shape).astype(np.float32))
    (out * w.to(DEV)).sum().backward()
    c = {k: v.requires_grad_(True) for k, v in _pt_args('cpu').items()}
    pc = torch.tensor(GOLD['pt_ptranslations']).requires_grad_(True)
    zc = torch.tensor(GOLD['pt_zooms']).requires_grad_(True)
    oc = pt._forward_elementwise(c['vertices'], scales=c['scales'], rotations=c['rotations'], translations=c['translations'],
                                 perspective_translations=pc, zooms=zc)
    (oc * w).sum().backward()
    for k in a:
        assert _rel(a[k].grad.cpu(), c[k].grad) <= 2e-5, (k, _rel(a[k].grad.cpu(), c[k].grad))
    assert _rel(pg.grad.cpu(), pc.grad) <= 2e-5 and _rel(zg.grad.cpu(), zc.grad) <= 2e-5


def test_derenderer3d_train_mode_render_uses_the_given_zoom_form():
    """Derenderer3d.render in train mode with _force_no_sample (the state scripts/main.py:422-423 puts the model in for
    the optimisation loop): pose tensors and vertices handed to the rasterizer against the element-wise CPU evaluation of
    the same blob."""
    from derender3d import TargetType
    from derender3d.models import Derenderer3d, ShapenetObj
    objs = [ShapenetObj(vertices=GOLD['template%d_vertices' % k], faces=GOLD['template%d_faces' % k]) for k in range(2)]
    got = {}
    for dev in (DEV, 'cpu'):
        m = Derenderer3d(mode=TargetType.extend, image_size=256, render_size=384, objs=objs).to(dev).train()
        m._force_no_sample = True
        m.batched = dev != 'cpu'
        calls = []

        def record(vertices, faces, normal=True, depth=True, calls=calls):
            calls.append(vertices.detach().cpu())
            n = vertices.shape[0]
            z = torch.zeros(n, 1, 8, 8, device=vertices.device)
            return z, torch.zeros(n, 3, 8, 8, device=vertices.device), z
        m.renderer.render_maps = record
        blob = {k[4:]: torch.tensor(GOLD[k], device=dev) for k in GOLD.files if k.startswith('blob_')}
        with torch.no_grad():
            res = m.render(blob)
        got[dev] = (res, calls)
    nverts = GOLD['render_nverts']
    for k in ('_thetas', '_alphas', '_rotations', '_scales', '_depths', '_center2ds', '_translations', '_zooms'):
        np.testing.assert_allclose(got[DEV][0][k].cpu().numpy(), got['cpu'][0][k].numpy(), rtol=2e-6, atol=2e-6, err_msg=k)
    vb = got[DEV][1][0]
    for i, vc in enumerate(got['cpu'][1]):
        nv = int(nverts[i])
        np.testing.assert_allclose(vb[i, :nv].numpy(), vc[0].numpy(), rtol=2e-5, atol=2e-6 * float(vc.abs().max()))


@pytest.mark.parametrize('batched', [True, False])
def test_derenderer3d_render_matches_reference_golden(batched):
    """Derenderer3d.render on cuda:0 (batched: FFDBank + fused transform; loop: the reference-shaped per-object path)
    for the golden blob: every pose tensor, the zooms, and the vertices handed to the rasterizer equal what the
    reference's render() produced (models/__init__.py:94-250, eval mode)."""
    from derender3d import TargetType
    from derender3d.models import Derenderer3d, ShapenetObj
    from derender3d.models.renderer import Renderer
    objs = [ShapenetObj(vertices=GOLD['template%d_vertices' % k], faces=GOLD['template%d_faces' % k]) for k in range(2)]
    m = Derenderer3d(mode=TargetType.extend, image_size=256, render_size=384, objs=objs).to(DEV).eval()
    m.batched = batched
    assert type(m.renderer) is Renderer
    calls = []

    def record(vertices, faces, normal=True, depth=True):
        calls.append((vertices.detach().cpu().numpy().copy(), m.renderer.viewing_angle, faces.detach().cpu().numpy().copy()))
        n = vertices.shape[0]
        z = torch.zeros(n, 1, 8, 8, device=vertices.device)
        return z, torch.zeros(n, 3, 8, 8, device=vertices.device), z

    m.renderer.render_maps = record      # instance attribute: the class (and with it the batched dispatch) is unchanged
    blob = {k[4:]: torch.tensor(GOLD[k], device=DEV) for k in GOLD.files if k.startswith('blob_')}
    with torch.no_grad():
        res = m.render(blob)
    for k in ('_thetas', '_alphas', '_rotations', '_scales', '_depths', '_center2ds', '_translations',
              '_class_log_probs', '_zooms'):
        assert res[k].device.type == torch.device(DEV).type
        np.testing.assert_allclose(res[k].cpu().numpy(), GOLD['render' + k], rtol=2e-6, atol=2e-6, err_msg=k)
    nverts = GOLD['render_nverts']
    n = len(nverts)
    if batched:
        assert len(calls) == 1
        v, ang, f = calls[0]
        per_obj = [(v[i], ang[i], f[i]) for i in range(n)]
    else:
        assert len(calls) == n
        per_obj = [(c[0][0], c[1], c[2][0]) for c in calls]
    classes = np.argmax(GOLD['blob_class_probs'], axis=1)
    for i, (v, ang, f) in enumerate(per_obj):
        nv = int(nverts[i])
        ref = GOLD['render_vertices'][i, :nv]
        # after zoom-to-fit z is O(10..100): relative gate with a floor at 1e-6 of the largest coordinate
        np.testing.assert_allclose(v[:nv], ref, rtol=2e-5, atol=2e-6 * float(np.abs(ref).max()), err_msg='object %d' % i)
        assert abs(float(ang) - GOLD['render_viewing_angles'][i]) < 1e-9
        tf = GOLD['template%d_faces' % classes[i]]
        assert np.array_equal(f[:tf.shape[0]], tf)


def test_derenderer3d_rendered_maps_match_oracle_on_reference_vertices():
    """The whole decode of Derenderer3d.render on cuda:0 (FFD -> PerspectiveTransform -> rasterizer) for the golden blob,
    against the renderer oracle applied to the vertices the REFERENCE's render() handed to its renderer: silhouette /
    normal / depth maps within 1e-4 abs (north_star's gate) on all but a 1e-4 fraction of the pixels (a 1e-6 relative
    vertex difference may move an edge pixel)."""
    from derender3d import TargetType
    from derender3d.models import Derenderer3d, ShapenetObj
    from oracle import nr_oracle as no
    objs = [ShapenetObj(vertices=GOLD['template%d_vertices' % k], faces=GOLD['template%d_faces' % k]) for k in range(2)]
    m = Derenderer3d(mode=TargetType.extend, image_size=256, render_size=384, objs=objs).to(DEV).eval()
    blob = {k[4:]: torch.tensor(GOLD[k], device=DEV) for k in GOLD.files if k.startswith('blob_')}
    with torch.no_grad():
        res = m.render(blob)            # render_size 384 like the golden: the zoom-to-fit depends on it
    classes = np.argmax(GOLD['blob_class_probs'], axis=1)
    for i in (0, 1):                    # one object per template class (classes of the golden blob: 0 1 0 0 0)
        nv = int(GOLD['render_nverts'][i])
        o = no.SDNRenderer(image_size=384, viewing_angle=float(GOLD['render_viewing_angles'][i]))
        v = torch.tensor(GOLD['render_vertices'][i:i + 1, :nv])
        f = torch.tensor(GOLD['template%d_faces' % classes[i]][None])
        ref = {'_masks': o(v, f, render_type=no.RenderType.Silhouette), '_normals': o(v, f, render_type=no.RenderType.Normal),
               '_depth_maps': o(v, f, render_type=no.RenderType.Depth)}
        for k, r in ref.items():
            d = (res[k][i:i + 1].cpu() - r).abs()
            assert float((d > 1e-4).float().mean()) <= 1e-4, (i, k, float(d.max()), float((d > 1e-4).float().mean()))


@pytest.mark.parametrize('n,m', [(16, 192), (5, 81), (1, 300)])
def test_ffd_coefficients_entry_point(n, m):
    """sdn_ffd_coefficients: out = base + x . M and its transpose x . M^T (the constraint map of FFD.constrain and its
    gradient direction) against float64, for sizes off the 256-thread grid as well."""
    from sdn_hip import check, lib, ptr, stream
    g = torch.Generator().manual_seed(n * 1000 + m)
    x = torch.randn(n, m, generator=g)
    M = torch.randn(m, m, generator=g) / m ** 0.5
    base = torch.randn(m, generator=g)
    xd, Md, bd = x.to(DEV), M.to(DEV), base.to(DEV)
    for transpose, with_base in ((0, True), (0, False), (1, False)):
        out = torch.full((n, m), float('nan'), device=DEV)
        check(lib().sdn_ffd_coefficients(ptr(xd), ptr(Md), ptr(bd) if with_base else None, n, m, transpose, ptr(out), stream()))
        want = x.double() @ (M.double().t() if transpose else M.double())
        if with_base:
            want = want + base.double()
        assert _rel(out.cpu(), want) <= 2e-6, (transpose, with_base, _rel(out.cpu(), want))
