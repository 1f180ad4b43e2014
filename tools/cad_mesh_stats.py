#!/usr/bin/env python3
"""Statistics of the ShapeNet CAD templates the reference loads (geometric/derender3d/models/__init__.py:18-59), measured
at the configs[1] pose -- ONLY statistics are stored (profiles/cad_mesh_stats.json): no vertex or face of the meshes.

Runs in the build container (needs /root/reference/geometric/assets).  Per mesh and pooled over the six present files:
  * triangle and vertex counts, share of exactly degenerate faces (repeated vertex index or zero 3-D area);
  * histogram of the projected area of the front-facing fill_back faces, in pixels of the internal S x S grid (S = 768),
    over log2 bins [2^-6 .. 2^12] (+ under / overflow);
  * depth complexity: sum of the projected areas of the front-facing faces / number of covered pixels (how many layers of
    geometry lie behind an average covered pixel: interior seats, engine parts, both sides of thin panels);
  * share of the covered pixels' faces that are "large" (> 64 px) -- panels -- vs "small" (< 4 px) -- wheels, grilles.
`sdn_hip.synth.cad_like` is fitted to these numbers (tests/test_cad_like.py compares its statistics with the stored ones);
`mesh_stats()` is the shared measuring function.

    python tools/cad_mesh_stats.py            # writes profiles/cad_mesh_stats.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

S = 768
BINS = np.arange(-6, 13)   # log2 of the area in pixels


def mesh_stats(verts, faces, render_size=384):
    """verts [V,3] normalised like ShapenetObj (unit extent, (z, y, -x) axes), faces [F,3] -> dict of statistics at the
    configs[1] pose (tests/util.posed_mesh: PerspectiveTransform, scale (3.9, 1.5, 1.6), theta 0.6, translation (2, 1, -12))."""
    from util import posed_mesh
    from oracle import nr_oracle as no
    import torch
    pv, ang = posed_mesh(verts, faces, render_size=render_size)
    # the renderer's camera: derender3d Renderer flips x, looks down -z, perspective with the viewing angle
    with torch.no_grad():
        v = torch.tensor(pv) * torch.tensor([-1., 1., 1.])
        v = no.look(v, torch.zeros(1, 3), torch.tensor([[0., 0., -1.]]), torch.tensor([[0., 1., 0.]]))
        v = no.perspective(v, angle=ang)[0].numpy().astype(np.float64)
    f = np.asarray(faces)
    tri = v[f]                                   # [F, 3, 3] NDC x, y + depth
    Sx = 2 * render_size
    px = 0.5 * (tri[..., :2] * Sx + Sx - 1)       # pixel coordinates (rasterize.py:138)
    a = px[:, 1] - px[:, 0]
    b = px[:, 2] - px[:, 0]
    signed = 0.5 * (a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0])
    area = np.abs(signed)                         # fill_back: one of the two windings is front-facing
    v3 = verts[f].astype(np.float64)
    a3 = 0.5 * np.linalg.norm(np.cross(v3[:, 1] - v3[:, 0], v3[:, 2] - v3[:, 0]), axis=1)
    degenerate = (f[:, 0] == f[:, 1]) | (f[:, 1] == f[:, 2]) | (f[:, 0] == f[:, 2]) | (a3 == 0)
    inside = (np.abs(tri[..., 0]).max(1) <= 1) & (np.abs(tri[..., 1]).max(1) <= 1)
    # faces the rasterizer sends down its band path (csrc/raster_fwd.hip face_margin_px: so thin that rounding lets pixels
    # beyond the bounding box pass the edge tests), evaluated on the fp32 NDC coordinates as the kernel does
    t32 = tri[..., :2].astype(np.float32)
    e0, e1, e2 = t32[:, 1] - t32[:, 0], t32[:, 2] - t32[:, 0], t32[:, 2] - t32[:, 1]
    l0, l1, l2 = (np.sqrt((e ** 2).sum(1)) for e in (e0, e1, e2))
    lmax, perim = np.maximum(l0, np.maximum(l1, l2)), l0 + l1 + l2
    area2 = np.abs(e0[:, 0] * e1[:, 1] - e1[:, 0] * e0[:, 1]) - np.float32(9.5367432e-7) * lmax * lmax
    delta = np.float32(9.5367432e-7) * (1 + np.abs(t32).max((1, 2)))
    with np.errstate(divide='ignore', invalid='ignore'):
        margin = 0.004 + 0.5 * Sx * delta + 0.5 * Sx * delta * perim * lmax / area2
    band = ~(area2 > 0) | (margin > 4.0)
    logs = np.log2(np.maximum(area[~degenerate], 1e-30))
    hist = np.histogram(np.clip(logs, BINS[0] - 1, BINS[-1] + 0.5), bins=np.concatenate([[BINS[0] - 1], BINS, [BINS[-1] + 1]]))[0]
    # covered pixels: the oracle's scanline kernel (fast) on the fill_back'ed faces
    r = no.SDNRenderer(image_size=render_size, viewing_angle=ang)
    r.raster_kw = {'unsafe': True}
    with torch.no_grad():
        m = r(torch.tensor(pv), torch.tensor(f[None].astype(np.int32)), render_type=no.RenderType.Silhouette)[0, 0].numpy()
    covered_ss = float(m.sum()) * 4.0             # 2x2 pooled coverage -> internal pixels
    return {
        'triangles': int(len(f)), 'vertices': int(len(verts)), 'degenerate_share': float(degenerate.mean()),
        'faces_in_view_share': float(inside.mean()), 'band_path_share': float(band.mean()),
        'area_log2_bins': [int(BINS[0]) - 1] + [int(x) for x in BINS],
        'area_hist_share': [float(x) / max(1, int(hist.sum())) for x in hist],
        'area_px_mean': float(area[~degenerate].mean()), 'area_px_median': float(np.median(area[~degenerate])),
        'area_px_p90': float(np.percentile(area[~degenerate], 90)), 'area_px_p99': float(np.percentile(area[~degenerate], 99)),
        'area_px_max': float(area.max()),
        'covered_pixels': covered_ss, 'depth_complexity': float(area.sum() / max(covered_ss, 1.0)),
        'area_share_small_lt4': float(area[area < 4].sum() / area.sum()),
        'area_share_large_gt64': float(area[area > 64].sum() / area.sum()),
        'count_share_small_lt4': float((area < 4).mean()), 'count_share_large_gt64': float((area > 64).mean()),
    }


def main():
    ref = os.environ.get('SDN_REFERENCE_ROOT', '/root/reference')
    assets = os.path.join(ref, 'geometric', 'assets')
    from derender3d.models import DEFAULT_OBJS, ShapenetObj
    per = {}
    for cls, oid in DEFAULT_OBJS:
        if not os.path.isdir(os.path.join(assets, cls, oid)):
            continue
        o = ShapenetObj(cls, oid, root_dir=assets)
        per['%s/%s' % (cls, oid[:8])] = mesh_stats(o.vertices.numpy(), o.faces.numpy())
    keys = [k for k, v in next(iter(per.values())).items() if isinstance(v, float)]
    pooled = {k: float(np.mean([m[k] for m in per.values()])) for k in keys}
    pooled['area_hist_share'] = [float(x) for x in np.mean([m['area_hist_share'] for m in per.values()], axis=0)]
    pooled['area_log2_bins'] = next(iter(per.values()))['area_log2_bins']
    pooled['triangles'] = float(np.mean([m['triangles'] for m in per.values()]))
    pooled['vertices'] = float(np.mean([m['vertices'] for m in per.values()]))
    out = {'source': 'the six OBJ files under geometric/assets of the reference checkout (statistics only)',
           'pose': 'configs[1]: tests/util.posed_mesh defaults, render_size 384 (S = 768)', 'per_mesh': per, 'pooled': pooled}
    path = os.path.join(ROOT, 'profiles', 'cad_mesh_stats.json')
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(pooled, indent=1))
    for k, m in per.items():
        print(k, m['triangles'], 'band %.4f' % m['band_path_share'], 'deg %.3f dc %.2f median %.2f p99 %.0f max %.0f small-count %.2f large-area %.2f'
              % (m['degenerate_share'], m['depth_complexity'], m['area_px_median'], m['area_px_p99'], m['area_px_max'],
                 m['count_share_small_lt4'], m['area_share_large_gt64']))


if __name__ == '__main__':
    main()
