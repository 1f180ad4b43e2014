"""HIP rasterizer (through the C ABI) against the CPU oracle and the reference-generated golden fixtures.

Bar (north_star): depth / normal maps within 1e-4 abs of the reference.  What is actually enforced here is
stricter: the maps are BIT-EQUAL to the oracle, and so is the silhouette / colour edge gradient (K5, one thread
per face in the reference's summation order); the atomically accumulated depth / texture gradients (K6, K7) are
compared at 1e-5 relative L2, since their summation order is unordered in the reference as well.
"""
import os

import numpy as np
import pytest
import torch

from oracle import nr_oracle as no
from util import biteq, random_soup

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'raster_golden.npz'))


def hip_rasterize(faces, textures, image_size, aa, flags, eps=1e-3, bg=(0.1, 0.2, 0.3), face_color=False,
                  eps_alpha=None, near=0.1, far=100):
    from sdn_hip import ops
    dev = torch.device('cuda:0')
    ft = torch.tensor(faces, device=dev, requires_grad=True)
    tt = torch.tensor(textures, device=dev, requires_grad=True) if flags[0] else None
    out = ops.RasterizeMaps.apply(ft, tt, image_size, aa, near, far, eps, bg, flags[0], flags[1], flags[2], eps_alpha,
                                  face_color)
    return ft, tt, out


def oracle_rasterize(faces, textures, image_size, aa, flags, eps=1e-3, bg=(0.1, 0.2, 0.3), near=0.1, far=100):
    fo = torch.tensor(faces, requires_grad=True)
    to = torch.tensor(textures, requires_grad=True) if flags[0] else None
    ref = no.rasterize_rgbad(fo, to, image_size, aa, near, far, eps, bg, flags[0], flags[1], flags[2])
    return fo, to, (ref['rgb'], ref['alpha'], ref['depth'])


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def run_pair(faces, textures, image_size, aa, flags, seed=0, face_color=False, serial=False, **kw):
    from sdn_hip import ops
    tex_o = textures
    if flags[0] and face_color:
        tex_o = np.ascontiguousarray(np.broadcast_to(textures[:, :, None, None, None, :],
                                                     textures.shape[:2] + (2, 2, 2, 3)))
    with ops.verification(serial_edges=serial):   # the backward pass uses what its forward call saw
        ft, tt, outs = hip_rasterize(faces, textures, image_size, aa, flags, face_color=face_color, **kw)
    fo, to, refs = oracle_rasterize(faces, tex_o, image_size, aa, flags, **kw)
    rng = np.random.default_rng(seed)
    lh = lo = 0
    for o, r in zip(outs, refs):
        if o is None:
            assert r is None
            continue
        assert biteq(o.detach().cpu().numpy(), r.detach().numpy())
        g = rng.normal(size=tuple(o.shape)).astype(np.float32)
        lh = lh + (o * torch.tensor(g, device=o.device)).sum()
        lo = lo + (r * torch.tensor(g)).sum()
    lh.backward()
    lo.backward()
    gh, go = ft.grad.cpu().numpy(), fo.grad.numpy()
    if not flags[2] and serial:
        assert biteq(gh, go), 'serial edge gradient must be bit-exact (max diff %g)' % np.abs(gh - go).max()
    else:
        assert rel_l2(gh, go) < 1e-5
    if flags[0]:
        gth, gto = tt.grad.cpu().numpy(), to.grad.numpy()
        if face_color:
            gto = gto.reshape(gto.shape[0], gto.shape[1], 8, 3).sum(2)
        assert rel_l2(gth, gto) < 1e-5


@pytest.mark.parametrize('bs,nf,is_,scale', [(1, 40, 32, 0.3), (2, 300, 48, 0.1), (1, 3000, 128, 0.03),
                                             (1, 500, 50, 0.2), (3, 64, 17, 0.4)])
@pytest.mark.parametrize('flags,aa', [((False, True, False), True), ((False, False, True), True),
                                      ((False, True, True), False)])
@pytest.mark.parametrize('serial', [False, True])
def test_soup_alpha_depth(bs, nf, is_, scale, flags, aa, serial):
    rng = np.random.default_rng(bs * 1000 + nf)
    run_pair(random_soup(rng, bs, nf, scale), None, is_, aa, flags, eps=1e-4, bg=None, serial=serial)


@pytest.mark.parametrize('nf,is_,scale,ts', [(40, 32, 0.3, 2), (3000, 128, 0.03, 2), (500, 50, 0.2, 4)])
@pytest.mark.parametrize('flags', [(True, False, False), (True, True, True)])
def test_soup_textured(nf, is_, scale, ts, flags):
    rng = np.random.default_rng(nf)
    faces = random_soup(rng, 1, nf, scale)
    tex = rng.uniform(0, 1, (1, nf, ts, ts, ts, 3)).astype(np.float32)
    run_pair(faces, tex, is_, True, flags)
    run_pair(faces, tex, is_, True, flags, serial=True)


def test_list_overflow_path_gives_identical_maps():
    """The tile kernel has two ways to find its faces (its own list / streaming every face's tile box)."""
    from sdn_hip import ops
    rng = np.random.default_rng(31)
    faces = random_soup(rng, 2, 5000, 0.05)
    _, _, (_, a1, d1) = hip_rasterize(faces, None, 160, True, (False, True, True), eps=1e-4, bg=None)
    with ops.verification(stream_faces=True):
        _, _, (_, a2, d2) = hip_rasterize(faces, None, 160, True, (False, True, True), eps=1e-4, bg=None)
        run_pair(faces[:1, :800], None, 48, True, (False, True, True), eps=1e-4, bg=None)
    assert torch.equal(a1, a2) and torch.equal(d1, d2)


def test_chunk_overflow_falls_back_to_the_serial_walk():
    """K5's chunk plan has room for 4 * faces + 65536 chunks; 300 triangles that span a 512-pixel image need ~115k, so
    part of them take the literal serial walk inside k_edge_reduce and the slots they reserved are marked invalid.  The
    gradient must still match the oracle (1e-5 relative L2, like every wave-mode comparison)."""
    rng = np.random.default_rng(77)
    faces = random_soup(rng, 1, 300, 0.9)
    run_pair(faces, None, 256, True, (False, True, False), eps=1e-4, bg=None)


def test_soup_face_color():
    rng = np.random.default_rng(9)
    faces = random_soup(rng, 1, 800, 0.08)
    col = rng.uniform(-1, 1, (1, 800, 3)).astype(np.float32)
    run_pair(faces, col, 64, True, (True, True, True), face_color=True)


def test_near_far_window():
    rng = np.random.default_rng(13)
    faces = random_soup(rng, 1, 300, 0.3, zlo=0.05, zhi=3.0)
    for near, far in ((0.5, 2.0), (0.3, 1.7)):
        run_pair(faces, None, 40, True, (False, True, True), eps=1e-4, bg=None, near=near, far=far)


def test_degenerate_and_sliver_faces_cover_what_the_reference_covers():
    g = lambda k: GOLD['slivers/' + k]
    ft, tt, (rgb, alpha, depth) = hip_rasterize(g('faces'), g('textures'), int(g('image_size')), False,
                                                (True, True, True))
    # golden maps are un-flipped S x S (Rasterize.forward_gpu); ours are flipped (rasterize_rgbad)
    assert biteq(alpha.detach().cpu().numpy()[:, ::-1], g('alpha_map'))
    assert biteq(depth.detach().cpu().numpy()[:, ::-1], g('depth_map'))
    assert biteq(rgb.detach().cpu().numpy()[:, :, ::-1].transpose(0, 2, 3, 1), g('rgb_map'))


@pytest.mark.parametrize('name', ['soup_small', 'soup_mid', 'slivers', 'cube'])
def test_reference_golden(name):
    """Forward maps and gradients produced by the reference's own kernels (tests/golden/make_raster_golden.py)."""
    g = lambda k: GOLD[name + '/' + k]
    is_ = int(g('image_size'))
    ft, tt, (rgb, alpha, depth) = hip_rasterize(g('faces'), g('textures'), is_, False, (True, True, True))
    assert biteq(alpha.detach().cpu().numpy()[:, ::-1], g('alpha_map'))
    assert biteq(depth.detach().cpu().numpy()[:, ::-1], g('depth_map'))
    assert biteq(rgb.detach().cpu().numpy()[:, :, ::-1].transpose(0, 2, 3, 1), g('rgb_map'))
    dev = rgb.device
    loss = (rgb * torch.tensor(np.ascontiguousarray(g('g_rgb').transpose(0, 3, 1, 2)[:, :, ::-1]), device=dev)).sum() \
        + (alpha * torch.tensor(np.ascontiguousarray(g('g_alpha')[:, ::-1]), device=dev)).sum() \
        + (depth * torch.tensor(np.ascontiguousarray(g('g_depth')[:, ::-1]), device=dev)).sum()
    loss.backward()
    assert rel_l2(ft.grad.cpu().numpy(), g('grad_faces')) < 1e-5
    assert rel_l2(tt.grad.cpu().numpy(), g('grad_textures')) < 1e-5
    # silhouette-only gradient: bit-exact in the reference's serial order, re-association error only otherwise
    from sdn_hip import ops
    for serial in (True, False):
        with ops.verification(serial_edges=serial):
            ft2, _, (_, alpha2, _) = hip_rasterize(g('faces'), None, is_, False, (False, True, False), eps=1e-4, bg=None)
        (alpha2 * torch.tensor(np.ascontiguousarray(g('g_alpha')[:, ::-1]), device=dev)).sum().backward()
        if serial:
            assert biteq(ft2.grad.cpu().numpy(), g('grad_faces_alpha_only'))
        else:
            assert rel_l2(ft2.grad.cpu().numpy(), g('grad_faces_alpha_only')) < 1e-6


def test_rasterize_class_returns_raw_maps():
    import neural_renderer as nr
    g = lambda k: GOLD['soup_small/' + k]
    dev = torch.device('cuda:0')
    R = nr.Rasterize(int(g('image_size')), 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, True)
    rgb, alpha, depth = R(torch.tensor(g('faces'), device=dev), torch.tensor(g('textures'), device=dev))
    assert biteq(rgb.cpu().numpy(), g('rgb_map'))
    assert biteq(alpha.cpu().numpy(), g('alpha_map'))
    assert biteq(depth.cpu().numpy(), g('depth_map'))


def test_determinism_and_batch_independence():
    """Size-independent properties at the benchmark's internal resolution (768^2): two runs are bit-identical, and
    a batch renders each element exactly as it renders alone."""
    rng = np.random.default_rng(21)
    faces = random_soup(rng, 3, 20000, 0.02)
    _, _, (_, a1, d1) = hip_rasterize(faces, None, 384, True, (False, True, True), eps=1e-4, bg=None)
    _, _, (_, a2, d2) = hip_rasterize(faces, None, 384, True, (False, True, True), eps=1e-4, bg=None)
    assert torch.equal(a1, a2) and torch.equal(d1, d2)
    for b in range(3):
        _, _, (_, ab, db) = hip_rasterize(faces[b:b + 1], None, 384, True, (False, True, True), eps=1e-4, bg=None)
        assert torch.equal(ab[0], a1[b]) and torch.equal(db[0], d1[b])
    # alpha is a 2x2 box filter of a 0/1 map; depth is bounded by near/far
    vals = torch.unique(a1)
    assert set(vals.tolist()) <= {0.0, 0.25, 0.5, 0.75, 1.0}
    assert float(d1.max()) <= 100.0 and float(d1.min()) > 0.1
    # fill ordering independence: permuting faces changes nothing but the tie-break among equal depths
    perm = rng.permutation(20000)
    _, _, (_, ap, dp) = hip_rasterize(faces[:, perm], None, 384, True, (False, True, True), eps=1e-4, bg=None)
    assert torch.equal(ap, a1) and torch.equal(dp, d1)


def test_empty_scene_and_offscreen_faces():
    faces = np.zeros((1, 5, 3, 3), np.float32)
    faces[..., 2] = 1.0
    faces[0, :, :, 0] += 5.0  # all off screen (and zero-area)
    _, _, (_, alpha, depth) = hip_rasterize(faces, None, 32, True, (False, True, True), eps=1e-4, bg=None)
    assert float(alpha.abs().max()) == 0.0 and torch.all(depth == 100.0)


def test_argument_errors():
    from sdn_hip import ops
    dev = torch.device('cuda:0')
    f = torch.zeros(1, 4, 3, 3, device=dev)
    with pytest.raises(Exception):
        ops.RasterizeMaps.apply(f, None, 16, True, 0.1, 100, 1e-4, None, False, False, False, None, False)
    with pytest.raises(ValueError):
        ops.RasterizeMaps.apply(f[0], None, 16, True, 0.1, 100, 1e-4, None, False, True, False, None, False)
    with pytest.raises(TypeError):
        ops.RasterizeMaps.apply(f.double(), None, 16, True, 0.1, 100, 1e-4, None, False, True, False, None, False)
    with pytest.raises(ValueError):
        ops.RasterizeMaps.apply(f, torch.zeros(1, 4, 1, 1, 1, 3, device=dev), 16, True, 0.1, 100, 1e-4, (0, 0, 0),
                                True, False, False, None, False)


def test_hierarchical_depth_cull_settings_draw_identical_maps():
    """SDN_RASTER_HIZ (read once per process): 0 = off, 1 = in every tile, 2 = in the tiles with long lists (the default since r06).
    The cull drops whole faces that lie behind every 8 x 8 block their box touches -- it must never change a pixel.  A CAD-like
    template (depth complexity ~8, tile lists of several hundred entries: the long-list path) and a fill_back'ed soup, rendered in
    one subprocess per setting; face-index-exact maps are compared through their raw bytes."""
    import hashlib
    import subprocess
    import sys
    code = r'''
import hashlib, sys
sys.path[:0] = %r
import numpy as np, torch
from derender3d.models.renderer import Renderer
from sdn_hip import synth
from util import posed_mesh
h = hashlib.sha256()
for seed, n in ((3, 46000), (4, 20000)):
    v, f = synth.cad_like(n, seed=seed)
    pv, ang = posed_mesh(v, f, render_size=256)
    r = Renderer(image_size=256)
    r.viewing_angle = [ang]
    m, nrm, d = r.render_maps(torch.tensor(pv, device='cuda:0'), torch.tensor(f[None].astype(np.int32), device='cuda:0'))
    for t in (m, nrm, d):
        h.update(t.detach().cpu().numpy().tobytes())
print('MAPS', h.hexdigest())
''' % ([p for p in sys.path if p],)
    digests = {}
    for setting in ('0', '1', '2'):
        env = dict(os.environ, SDN_RASTER_HIZ=setting)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('MAPS ')]
        assert r.returncode == 0 and line, r.stdout[-500:] + r.stderr[-1500:]
        digests[setting] = line[0]
    assert digests['0'] == digests['1'] == digests['2'], digests
