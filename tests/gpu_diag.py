"""Verbose HIP-vs-oracle comparison, meant for a gpurun call:  python tests/gpu_diag.py [quick|full]
Writes a log of every compared quantity to stdout (and mismatching pixel dumps to gpurun_out/diag_*.npz)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
    sys.path.insert(0, p)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import raster_np as rn  # noqa: E402
from oracle import nr_oracle as no  # noqa: E402
import sdn_hip  # noqa: E402
from sdn_hip import ops, synth  # noqa: E402
from util import random_soup, biteq, posed_mesh  # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
dev = torch.device('cuda:0')


def cmp(name, a, b, tol=None):
    a = np.asarray(a)
    b = np.asarray(b)
    if biteq(a, b):
        print('  %-22s BITEQ' % name)
        return True
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    nbad = int((d > (tol or 0)).sum())
    print('  %-22s max|d|=%.3e  n(|d|>%g)=%d of %d  (NaN a/b: %d/%d)' % (
        name, np.nanmax(d), tol or 0, nbad, d.size, np.isnan(a).sum(), np.isnan(b).sum()))
    return nbad == 0


def raster_case(tag, faces, textures, image_size, aa, flags, eps=1e-3, bg=(0.1, 0.2, 0.3), face_color=False,
                check_bwd=True, seed=0):
    rr, ra, rd = flags
    print('[%s] bs=%d nf=%d is=%d aa=%d rgb=%d alpha=%d depth=%d face_color=%d' % (
        tag, faces.shape[0], faces.shape[1], image_size, aa, rr, ra, rd, face_color))
    ft = torch.tensor(faces, device=dev, requires_grad=True)
    tt = None
    tex_o = textures
    if rr:
        tt = torch.tensor(textures, device=dev, requires_grad=True)
        if face_color:
            tex_o = np.ascontiguousarray(np.broadcast_to(textures[:, :, None, None, None, :],
                                                         textures.shape[:2] + (2, 2, 2, 3)))
    t0 = time.time()
    rgb, alpha, depth = ops.RasterizeMaps.apply(ft, tt, image_size, aa, 0.1, 100, eps, bg, rr, ra, rd, None,
                                                face_color)
    torch.cuda.synchronize()
    t1 = time.time()
    fo = torch.tensor(faces, requires_grad=True)
    to = torch.tensor(tex_o, requires_grad=True) if rr else None
    ref = no.rasterize_rgbad(fo, to, image_size, aa, 0.1, 100, eps, bg, rr, ra, rd)
    t2 = time.time()
    print('  hip %.1f ms (incl. launch+sync), oracle %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    ok = True
    if rr:
        ok &= cmp('rgb', rgb.detach().cpu().numpy(), ref['rgb'].detach().numpy())
    if ra:
        ok &= cmp('alpha', alpha.detach().cpu().numpy(), ref['alpha'].detach().numpy())
    if rd:
        ok &= cmp('depth', depth.detach().cpu().numpy(), ref['depth'].detach().numpy())
    if check_bwd:
        rng = np.random.default_rng(seed + 17)
        loss_h = 0
        loss_o = 0
        if rr:
            g = rng.normal(size=tuple(rgb.shape)).astype(np.float32)
            loss_h = loss_h + (rgb * torch.tensor(g, device=dev)).sum()
            loss_o = loss_o + (ref['rgb'] * torch.tensor(g)).sum()
        if ra:
            g = rng.normal(size=tuple(alpha.shape)).astype(np.float32)
            loss_h = loss_h + (alpha * torch.tensor(g, device=dev)).sum()
            loss_o = loss_o + (ref['alpha'] * torch.tensor(g)).sum()
        if rd:
            g = rng.normal(size=tuple(depth.shape)).astype(np.float32)
            loss_h = loss_h + (depth * torch.tensor(g, device=dev)).sum()
            loss_o = loss_o + (ref['depth'] * torch.tensor(g)).sum()
        loss_h.backward()
        loss_o.backward()
        gh = ft.grad.cpu().numpy()
        go = fo.grad.numpy()
        if not biteq(gh, go):
            rel = np.linalg.norm(gh.astype(np.float64) - go) / max(np.linalg.norm(go.astype(np.float64)), 1e-30)
            print('  grad_faces             rel L2 = %.3e   max|d| = %.3e   max|ref| = %.3e' % (
                rel, np.abs(gh - go).max(), np.abs(go).max()))
            ok &= rel < 1e-4
        else:
            print('  grad_faces             BITEQ')
        if rr:
            gth = tt.grad.cpu().numpy()
            gto = to.grad.numpy()
            if face_color:
                gto = gto.reshape(gto.shape[0], gto.shape[1], 8, 3).sum(2)
            rel = np.linalg.norm(gth.astype(np.float64) - gto) / max(np.linalg.norm(gto.astype(np.float64)), 1e-30)
            print('  grad_textures          rel L2 = %.3e' % rel)
            ok &= rel < 1e-4
    print('  => %s' % ('OK' if ok else 'MISMATCH'))
    return ok


def renderer_case(tag, verts, faces, angle, render_size, want_grad=True, normal_grad=True):
    from derender3d.models.renderer import Renderer, RenderType
    print('[%s] V=%d F0=%d R=%d angle=%.4f' % (tag, verts.shape[1], faces.shape[0], render_size, angle))
    r = Renderer(image_size=render_size)
    r.viewing_angle = angle
    vt = torch.tensor(verts, device=dev, requires_grad=True)
    fi = torch.tensor(faces[None], device=dev)
    t0 = time.time()
    m, n, d = r.render_maps(vt, fi)
    torch.cuda.synchronize()
    t1 = time.time()
    o = no.SDNRenderer(image_size=render_size, viewing_angle=angle)
    vo = torch.tensor(verts, requires_grad=True)
    fo = torch.tensor(faces[None])
    mo = o(vo, fo, render_type=no.RenderType.Silhouette)
    nno = o(vo, fo, render_type=no.RenderType.Normal)
    do = o(vo, fo, render_type=no.RenderType.Depth)
    t2 = time.time()
    print('  hip fused %.1f ms, oracle 3 renders %.1f s' % ((t1 - t0) * 1e3, t2 - t1))
    ok = True
    ok &= cmp('mask', m.detach().cpu().numpy(), mo.detach().numpy(), 1e-4)
    ok &= cmp('normal', n.detach().cpu().numpy(), nno.detach().numpy(), 1e-4)
    ok &= cmp('depth', d.detach().cpu().numpy(), do.detach().numpy(), 1e-4)
    # separate calls must equal the fused ones
    m2 = r(vt, fi, render_type=RenderType.Silhouette)
    n2 = r(vt, fi, render_type=RenderType.Normal)
    d2 = r(vt, fi, render_type=RenderType.Depth)
    ok &= cmp('mask  fused==single', m.detach().cpu().numpy(), m2.detach().cpu().numpy())
    ok &= cmp('normal fused==single', n.detach().cpu().numpy(), n2.detach().cpu().numpy())
    ok &= cmp('depth fused==single', d.detach().cpu().numpy(), d2.detach().cpu().numpy())
    if want_grad:
        rng = np.random.default_rng(5)
        gm = rng.uniform(-1, 1, tuple(m.shape)).astype(np.float32)
        gn = rng.uniform(-1, 1, tuple(n.shape)).astype(np.float32)
        gd = rng.uniform(-1, 1, tuple(d.shape)).astype(np.float32)
        cases = [('mask', ((m, mo, gm),)), ('depth', ((d, do, gd),))]
        if normal_grad:
            cases += [('normal', ((n, nno, gn),)), ('all', ((m, mo, gm), (n, nno, gn), (d, do, gd)))]
        for name, terms in cases:
            vt.grad = None
            vo.grad = None
            lh = sum((a * torch.tensor(g, device=dev)).sum() for a, _, g in terms)
            lo = sum((b * torch.tensor(g)).sum() for _, b, g in terms)
            lh.backward(retain_graph=True)
            lo.backward(retain_graph=True)
            gh = vt.grad.cpu().numpy().astype(np.float64)
            go = vo.grad.numpy().astype(np.float64)
            rel = np.linalg.norm(gh - go) / max(np.linalg.norm(go), 1e-30)
            print('  grad_vertices[%s]   rel L2 = %.3e  |ref| = %.3e  nan hip/ref = %d/%d' % (
                name, rel, np.linalg.norm(go), np.isnan(gh).sum(), np.isnan(go).sum()))
            ok &= rel < 1e-3
    print('  => %s' % ('OK' if ok else 'MISMATCH'))
    return ok


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'quick'
    print('device:', torch.cuda.get_device_name(0), ' lib:', sdn_hip.LIB_PATH, ' oracle threads:', rn.num_threads())
    rng = np.random.default_rng(1)
    results = []
    # 1. raster core on random soups (bit-exact expected for maps and for the alpha-only edge gradient)
    for (bs, nf, is_, scale) in [(1, 40, 32, 0.3), (2, 300, 48, 0.1), (1, 3000, 128, 0.03), (1, 500, 50, 0.2)]:
        faces = random_soup(rng, bs, nf, scale)
        tex = rng.uniform(0, 1, (bs, nf, 2, 2, 2, 3)).astype(np.float32)
        col = rng.uniform(-1, 1, (bs, nf, 3)).astype(np.float32)
        results.append(raster_case('soup-alpha', faces, None, is_, True, (False, True, False)))
        results.append(raster_case('soup-depth', faces, None, is_, True, (False, False, True)))
        results.append(raster_case('soup-noaa-ad', faces, None, is_, False, (False, True, True)))
        if bs == 1:
            results.append(raster_case('soup-rgb', faces, tex, is_, True, (True, False, False)))
            results.append(raster_case('soup-rgbad', faces, tex, is_, True, (True, True, True)))
            results.append(raster_case('soup-facecolor', faces, col, is_, True, (True, True, True), face_color=True))
    # 2. full renderer on meshes
    v, f = synth.cube()
    pv, ang = posed_mesh(v, f, theta=0.5, scale=(1, 1, 1), translation=(0.3, 0.2, -3.0), render_size=128)
    results.append(renderer_case('cube', pv, f, ang, 128))
    v, f = synth.car_like(2000, seed=1)
    pv, ang = posed_mesh(v, f, render_size=64)
    results.append(renderer_case('car2k-64', pv, f, ang, 64, normal_grad=False))
    v, f = synth.car_like(2000, seed=1, degenerate=0)
    pv, ang = posed_mesh(v, f, render_size=64)
    results.append(renderer_case('car2k-64-nodegen', pv, f, ang, 64))
    if mode == 'full':
        v, f = synth.car_like(45000, seed=2)
        pv, ang = posed_mesh(v, f)
        results.append(renderer_case('car45k-384', pv, f, ang, 384, normal_grad=False))
    print('SUMMARY: %d/%d cases OK' % (sum(results), len(results)))
    return 0 if all(results) else 1


if __name__ == '__main__':
    sys.exit(main())
