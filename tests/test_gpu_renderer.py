"""derender3d Renderer / neural_renderer.Renderer on the GPU against the torch-CPU oracle of the reference graph."""
import numpy as np
import pytest
import torch

from oracle import nr_oracle as no
from sdn_hip import synth
from util import biteq, posed_mesh

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def both_renderers(pv, f, ang, R):
    from derender3d.models.renderer import Renderer
    r = Renderer(image_size=R)
    r.viewing_angle = ang
    o = no.SDNRenderer(image_size=R, viewing_angle=ang)
    vt = torch.tensor(pv, device=DEV, requires_grad=True)
    vo = torch.tensor(pv, requires_grad=True)
    return r, o, vt, vo, torch.tensor(f[None], device=DEV), torch.tensor(f[None])


def close_maps(h, o, tol=1e-4, max_bad_frac=1e-5):
    """north_star gate: 1e-4 abs; a differing edge pixel (1-ulp vertex differences) is tolerated at <= 1e-5 of pixels."""
    d = np.abs(h.astype(np.float64) - o.astype(np.float64))
    bad = (d > tol).sum()
    assert bad <= max_bad_frac * d.size, '%d of %d pixels differ by more than %g (max %g)' % (bad, d.size, tol, d.max())


@pytest.mark.parametrize('mesh,R', [('cube', 128), ('car2k', 64), ('sphere', 96)])
def test_three_maps_forward_and_backward(mesh, R):
    if mesh == 'cube':
        v, f = synth.cube()
        pv, ang = posed_mesh(v, f, theta=0.5, scale=(1, 1, 1), translation=(0.3, 0.2, -3.0), render_size=R)
    elif mesh == 'car2k':
        v, f = synth.car_like(2000, seed=1, degenerate=0)
        pv, ang = posed_mesh(v, f, render_size=R)
    else:
        v, f = synth.uv_sphere(24, 32)
        pv, ang = posed_mesh(v * 0.5, f, theta=1.1, scale=(2, 2, 2), translation=(-1.0, 0.5, -9.0), render_size=R)
    r, o, vt, vo, fi, fo = both_renderers(pv, f, ang, R)
    m, n, d = r.render_maps(vt, fi)
    mo = o(vo, fo, render_type=no.RenderType.Silhouette)
    nno = o(vo, fo, render_type=no.RenderType.Normal)
    do = o(vo, fo, render_type=no.RenderType.Depth)
    assert m.shape == (1, 1, R, R) and n.shape == (1, 3, R, R) and d.shape == (1, 1, R, R)
    close_maps(m.detach().cpu().numpy(), mo.detach().numpy())
    close_maps(n.detach().cpu().numpy(), nno.detach().numpy())
    close_maps(d.detach().cpu().numpy(), do.detach().numpy())
    rng = np.random.default_rng(5)
    for (h, ref) in ((m, mo), (n, nno), (d, do)):
        g = rng.uniform(-1, 1, tuple(h.shape)).astype(np.float32)
        vt.grad = None
        vo.grad = None
        (h * torch.tensor(g, device=DEV)).sum().backward(retain_graph=True)
        (ref * torch.tensor(g)).sum().backward(retain_graph=True)
        assert rel_l2(vt.grad.cpu().numpy(), vo.grad.numpy()) < 1e-4  # 1e-4 rel L2 on gradients


def test_unused_maps_launch_no_backward_kernels():
    """A silhouette-only loss (the test-time optimisation, scripts/main.py:445-453) leaves the normal map without a
    gradient: the face-normal and its vertex-gather backward must not run at all (autograd would otherwise hand them
    zero tensors and launch both); a loss on the normal map runs each exactly once.  Checked on the composition of the
    separate Functions (launch counts) and on the one-call path (sdn_render_maps_bwd receives NULL for the maps without a
    gradient, which is what makes it skip the colour pass and the normal branch, csrc/raster_maps.hip)."""
    import sdn_hip
    v, f = synth.car_like(2000, seed=2)
    pv, ang = posed_mesh(v, f, render_size=64)
    r, _, vt, _, fi, _ = both_renderers(pv, f, ang, 64)
    L = sdn_hip.lib()
    calls = {'n': 0, 'g': 0, 'maps': []}
    real_n, real_g, real_m = L.sdn_face_normals_bwd, L.sdn_gather_faces_bwd, L.sdn_render_maps_bwd

    def count_n(*a):
        calls['n'] += 1
        return real_n(*a)

    def count_g(*a):
        calls['g'] += 1
        return real_g(*a)

    def see_maps(*a):
        calls['maps'].append(tuple(bool(getattr(p, 'value', p)) for p in a[18:21]))   # g_alpha, g_normal, g_depth (behind bg)
        return real_m(*a)
    L.sdn_face_normals_bwd, L.sdn_gather_faces_bwd, L.sdn_render_maps_bwd = count_n, count_g, see_maps
    try:
        m, n, d = r.render_maps_composed(vt, fi)
        (m ** 2).sum().backward(retain_graph=True)
        assert (calls['n'], calls['g']) == (0, 1), calls          # only the projected vertices' gather
        g_mask = vt.grad.clone()
        assert float(g_mask.abs().max()) > 0
        (n ** 2).sum().backward()
        assert (calls['n'], calls['g']) == (1, 3), calls
        assert float((vt.grad - g_mask).abs().max()) > 0
        assert not calls['maps']
        vt.grad = None
        m, n, d = r.render_maps(vt, fi)
        (m ** 2).sum().backward(retain_graph=True)
        assert calls['maps'] == [(True, False, False)], calls
        assert float((vt.grad - g_mask).norm() / g_mask.norm()) <= 1e-6
        (n ** 2).sum().backward()
        assert calls['maps'][1] == (False, True, False), calls
        assert (calls['n'], calls['g']) == (1, 3)                 # the one-call path does not go through the Python bindings
    finally:
        L.sdn_face_normals_bwd, L.sdn_gather_faces_bwd, L.sdn_render_maps_bwd = real_n, real_g, real_m


def test_single_calls_equal_fused_call():
    from derender3d.models.renderer import RenderType
    v, f = synth.car_like(2000, seed=3)
    pv, ang = posed_mesh(v, f, render_size=64)
    r, _, vt, _, fi, _ = both_renderers(pv, f, ang, 64)
    m, n, d = r.render_maps(vt, fi)
    assert torch.equal(m, r(vt, fi, render_type=RenderType.Silhouette))
    assert torch.equal(n, r(vt, fi, render_type=RenderType.Normal))
    assert torch.equal(d, r(vt, fi, render_type=RenderType.Depth))
    # gradient of the fused call == sum of the gradients of the three separate calls
    gm, gn, gd = torch.rand_like(m), torch.rand_like(n), torch.rand_like(d)
    vt.grad = None
    ((m * gm).sum() + (n * gn).sum() + (d * gd).sum()).backward()
    g_fused = vt.grad.clone()
    vt.grad = None
    ((r(vt, fi, render_type=RenderType.Silhouette) * gm).sum() + (r(vt, fi, render_type=RenderType.Normal) * gn).sum()
     + (r(vt, fi, render_type=RenderType.Depth) * gd).sum()).backward()
    assert rel_l2(g_fused.cpu().numpy(), vt.grad.cpu().numpy()) < 1e-5
    assert torch.isfinite(g_fused).all()  # degenerate faces: finite (the reference's normalize backward gives NaN)


def test_rgb_render_with_lighting():
    from derender3d.models.renderer import RenderType
    v, f = synth.uv_sphere(12, 16)
    pv, ang = posed_mesh(v * 0.5, f, theta=0.3, scale=(2, 2, 2), translation=(0.5, 0.2, -8.0), render_size=48)
    r, o, vt, vo, fi, fo = both_renderers(pv, f, ang, 48)
    rng = np.random.default_rng(2)
    tex = rng.uniform(0, 1, (1, len(f), 2, 2, 2, 3)).astype(np.float32)
    th, to = torch.tensor(tex, device=DEV, requires_grad=True), torch.tensor(tex, requires_grad=True)
    ih = r(vt, fi, th, render_type=RenderType.RGB)
    io = o(vo, fo, to, render_type=no.RenderType.RGB)
    close_maps(ih.detach().cpu().numpy(), io.detach().numpy())
    g = rng.uniform(-1, 1, tuple(ih.shape)).astype(np.float32)
    (ih * torch.tensor(g, device=DEV)).sum().backward()
    (io * torch.tensor(g)).sum().backward()
    assert rel_l2(vt.grad.cpu().numpy(), vo.grad.numpy()) < 1e-4
    assert rel_l2(th.grad.cpu().numpy(), to.grad.numpy()) < 1e-4


def test_nr_renderer_look_at_defaults():
    import neural_renderer as nr
    v, f = synth.uv_sphere(10, 12)
    vt = torch.tensor(v[None] * 0.6, device=DEV)
    fi = torch.tensor(f[None], device=DEV)
    r = nr.Renderer()
    r.image_size = 64
    o = no.NRRenderer()
    o.image_size = 64
    sil = r.render_silhouettes(vt, fi)
    dep = r.render_depth(vt, fi)
    close_maps(sil.cpu().numpy(), o.render_silhouettes(torch.tensor(v[None] * 0.6), torch.tensor(f[None])).numpy())
    close_maps(dep.cpu().numpy(), o.render_depth(torch.tensor(v[None] * 0.6), torch.tensor(f[None])).numpy())


def test_camera_functions_match_oracle():
    import neural_renderer as nr
    rng = np.random.default_rng(1)
    v = rng.normal(size=(2, 50, 3)).astype(np.float32) + np.array([0, 0, 5], np.float32)
    eye = rng.normal(size=(2, 3)).astype(np.float32)
    dirs = rng.normal(size=(2, 3)).astype(np.float32)
    vt = torch.tensor(v, device=DEV, requires_grad=True)
    vo = torch.tensor(v, requires_grad=True)
    for fh, fo in ((lambda x: nr.look(x, torch.tensor(eye, device=DEV), torch.tensor(dirs, device=DEV)),
                    lambda x: no.look(x, eye, dirs)),
                   (lambda x: nr.look_at(x, torch.tensor(eye, device=DEV)), lambda x: no.look_at(x, eye)),
                   (lambda x: nr.perspective(x, 25.0), lambda x: no.perspective(x, 25.0))):
        a, b = fh(vt), fo(vo)
        assert biteq(a.detach().cpu().numpy(), b.detach().numpy())
        vt.grad = None
        vo.grad = None
        w = torch.linspace(-1, 1, a.numel()).reshape(a.shape)
        (a * w.to(DEV)).sum().backward()
        (b * w).sum().backward()
        assert rel_l2(vt.grad.cpu().numpy(), vo.grad.numpy()) < 1e-5
    faces = torch.tensor(rng.integers(0, 50, (2, 30, 3)).astype(np.int32))
    assert biteq(nr.vertices_to_faces(vt, faces.to(DEV)).detach().cpu().numpy(),
                 no.vertices_to_faces(vo, faces).detach().numpy())


@pytest.mark.timeout(900)
def test_numerics_gate_full_size_car():
    """BASELINE.json configs[1]: one car-sized mesh (42.9k triangles -> 85.7k faces with fill_back) at R = 384
    (768^2 internal), silhouette + normal + depth forward and the silhouette-loss backward, against the oracle."""
    v, f = synth.car_like(45000, seed=2)
    pv, ang = posed_mesh(v, f)
    r, o, vt, vo, fi, fo = both_renderers(pv, f, ang, 384)
    m, n, d = r.render_maps(vt, fi)
    mo = o(vo, fo, render_type=no.RenderType.Silhouette)
    close_maps(m.detach().cpu().numpy(), mo.detach().numpy())
    close_maps(d.detach().cpu().numpy(), o(vo, fo, render_type=no.RenderType.Depth).detach().numpy())
    close_maps(n.detach().cpu().numpy(), o(vo, fo, render_type=no.RenderType.Normal).detach().numpy())
    target = torch.zeros(1, 1, 384, 384)
    target[:, :, 100:300, 60:330] = 1
    ((m - target.to(DEV)) ** 2).mean().backward()
    ((mo - target) ** 2).mean().backward()
    assert rel_l2(vt.grad.cpu().numpy(), vo.grad.numpy()) < 1e-4


def test_numerics_gate_cad_statistics_mesh():
    """The same gate on a mesh with the STATISTICS of the ShapeNet CAD files the reference renders (synth.cad_like, fitted to
    profiles/cad_mesh_stats.json: depth complexity ~8, triangle areas from 1/100 pixel to thousands of pixels, slivers,
    degenerate faces) -- the regime car_like does not reach: interior layers make z-ties and near-ties common, big panels
    take the wave-shared path of the tile kernel, heavy tiles overflow less evenly.  R = 192 (S = 384) keeps the brute-force
    oracle (S^2 x faces inside tests) within seconds."""
    v, f = synth.cad_like(14000, seed=3)
    pv, ang = posed_mesh(v, f, render_size=192)
    r, o, vt, vo, fi, fo = both_renderers(pv, f, ang, 192)
    m, n, d = r.render_maps(vt, fi)
    mo = o(vo, fo, render_type=no.RenderType.Silhouette)
    close_maps(m.detach().cpu().numpy(), mo.detach().numpy())
    close_maps(d.detach().cpu().numpy(), o(vo, fo, render_type=no.RenderType.Depth).detach().numpy())
    close_maps(n.detach().cpu().numpy(), o(vo, fo, render_type=no.RenderType.Normal).detach().numpy())
    target = torch.zeros(1, 1, 192, 192)
    target[:, :, 50:150, 30:165] = 1
    ((m - target.to(DEV)) ** 2).mean().backward()
    ((mo - target) ** 2).mean().backward()
    assert rel_l2(vt.grad.cpu().numpy(), vo.grad.numpy()) < 1e-4


def test_fused_render_maps_equals_the_composed_functions():
    """Renderer.render_maps (sdn_render_maps_fwd / _bwd: one C call each way) against render_maps_composed (project, gather,
    face normals, rasterize as separate autograd Functions + the two x-flip multiplications): the same launchers in the
    same order, so the maps are equal bit for bit; the vertex gradients meet in float atomics in both (gather_bwd), hence
    1e-6 -- for all three maps, for the silhouette alone, and for a loss on the normal map only."""
    import numpy as np
    import torch
    from derender3d.models.renderer import Renderer
    from sdn_hip import synth
    from util import posed_mesh
    v, f = synth.car_like(3000, seed=5)
    pv, ang = posed_mesh(v, f, render_size=96)
    verts = np.concatenate([pv, pv * np.float32(1.03) + np.float32([0.2, -0.1, 0.0])]).astype(np.float32)
    r = Renderer(image_size=96)
    r.viewing_angle = [ang, ang]
    fi = torch.tensor(f[None], device='cuda:0').expand(2, -1, -1).contiguous()
    g = torch.Generator(device='cuda').manual_seed(3)
    wm, wn, wd = (torch.randn(s, generator=g, device='cuda') for s in ((2, 1, 96, 96), (2, 3, 96, 96), (2, 1, 96, 96)))

    def run(fn, which, **kw):
        vt = torch.tensor(verts, device='cuda:0', requires_grad=True)
        m, n, d = fn(vt, fi, **kw)
        loss = 0
        if 'm' in which:
            loss = loss + (m * wm).sum()
        if 'n' in which:
            loss = loss + (n * wn).sum()
        if 'd' in which:
            loss = loss + (d * wd).sum()
        loss.backward()
        return m, n, d, vt.grad
    # ('m', ...): only the silhouette is differentiated -- r06: the edge pass of sdn_render_maps_bwd then adds to the vertices itself
    # (rasterize_bwd_core's VertexSink: no face-gradient tensor, no k_gather_faces_bwd) while the composed path still gathers
    for which, kw in (('mnd', {}), ('m', {}), ('m', {'normal': False}), ('n', {}), ('md', {'normal': False}), ('d', {}), ('nd', {}),
                      ('d', {'normal': False})):
        a = run(r.render_maps, which, **kw)
        b = run(r.render_maps_composed, which, **kw)
        for x, y, name in zip(a[:3], b[:3], ('mask', 'normal', 'depth')):
            assert (x is None) == (y is None), (which, name)
            if x is not None:
                assert torch.equal(x, y), (which, name, float((x - y).abs().max()))
        rel = float((a[3] - b[3]).norm() / b[3].norm())
        assert rel <= 1e-6, (which, rel)
    # without fill_back (no reversed twin of every face: the vertex sink's other index path); both routes read the same defaults bag
    from derender3d.models import renderer as rmod
    bag = rmod._defaults()
    saved = bag.fill_back
    bag.fill_back = False
    try:
        for which in ('m', 'mnd'):
            a = run(r.render_maps, which)
            b = run(r.render_maps_composed, which)
            assert torch.equal(a[0], b[0]), which
            rel = float((a[3] - b[3]).norm() / b[3].norm())
            assert rel <= 1e-6, (which, 'fill_back off', rel)
    finally:
        bag.fill_back = saved
