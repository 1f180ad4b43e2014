"""GPU: what k_raster_tiles is given for the bench frame -- per-tile list lengths and candidate pixel counts, by mesh.

    python tools/tile_stats.py [car_like cad_like]

For each mesh the frame's 16 objects are decoded and projected exactly as bench.make_step does; the faces' pixel boxes are
recomputed here in torch (without the rounding margin of k_face_setup -- a statistics tool, not a checker) and binned into
the 32 x 32 tiles of the 768^2 internal image.  Printed: how the (face, tile) entries and the candidate pixel tests are
distributed over the tiles, which share belongs to 'large' clipped boxes (the wave-shared path) and the work counters of
the counting build."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
    if p not in sys.path:
        sys.path.insert(0, p)
TS = 32


def frame_faces(device, mesh):
    import bench
    from derender3d.models.renderer import Renderer
    bank, sizes, cls, params, targets, ptf = bench.build_scene(device, 0, mesh=mesh)
    n = bench.OBJECTS_PER_FRAME
    renderer = Renderer(image_size=bench.RENDER_SIZE)
    renderer.viewing_angle = [np.arctan(bench.RENDER_SIZE / (2.0 * bench.FOCAL)) / np.pi * 180] * n
    zoom_to = torch.full((n, 1), bench.RENDER_SIZE / (2.0 * bench.FOCAL), device=device)
    zeros = torch.zeros(n, 1, device=device)
    with torch.no_grad():
        verts, faces = bank.decode(params['ffd'], torch.tensor(cls, device=device, dtype=torch.int64))
        th = params['theta']
        rot = torch.cat([torch.cos(th / 2), zeros, torch.sin(th / 2), zeros], dim=1)
        tr = params['translation']
        verts, _ = ptf(verts, scales=torch.exp(params['log_scale']), rotations=rot, translations=tr,
                       perspective_translations=tr, zoom_tos=zoom_to)
        r, v = renderer._setup(verts)
        f9 = r.gather(r.project(v), faces)
    return f9, (bank, cls, params, targets, ptf)


def stats(f9, S):
    bs, nf = f9.shape[:2]
    x = (f9[..., 0] * 0.5 + 0.5) * S - 0.5          # pixel coordinates of the three vertices [bs, nf, 3]
    y = (f9[..., 1] * 0.5 + 0.5) * S - 0.5
    cross = (x[..., 1] - x[..., 0]) * (y[..., 2] - y[..., 0]) - (x[..., 2] - x[..., 0]) * (y[..., 1] - y[..., 0])
    front = cross != 0          # (both windings are in the list with fill_back; the kernel's back-face rule drops one)
    m = 0.01
    x0 = torch.ceil(x.min(-1)[0] - m).clamp(0, S - 1)
    x1 = torch.floor(x.max(-1)[0] + m).clamp(0, S - 1)
    y0 = torch.ceil(y.min(-1)[0] - m).clamp(0, S - 1)
    y1 = torch.floor(y.max(-1)[0] + m).clamp(0, S - 1)
    ok = front & (x1 >= x0) & (y1 >= y0) & (x.max(-1)[0] >= 0) & (y.max(-1)[0] >= 0) & (x.min(-1)[0] <= S - 1) & \
        (y.min(-1)[0] <= S - 1)
    ntx = S // TS
    entries = torch.zeros(bs, ntx * ntx, device=f9.device)
    cand = torch.zeros_like(entries)
    cand_big = torch.zeros_like(entries)
    n_big = torch.zeros_like(entries)
    for b in range(bs):
        k = ok[b].nonzero()[:, 0]
        bx0, bx1, by0, by1 = x0[b, k], x1[b, k], y0[b, k], y1[b, k]
        tx0, tx1, ty0, ty1 = (bx0 // TS).long(), (bx1 // TS).long(), (by0 // TS).long(), (by1 // TS).long()
        span = int(max((tx1 - tx0).max(), (ty1 - ty0).max())) + 1
        for dy in range(span):
            for dx in range(span):
                tx, ty = tx0 + dx, ty0 + dy
                use = (tx <= tx1) & (ty <= ty1)
                if not bool(use.any()):
                    continue
                cx0 = torch.maximum(bx0, (tx * TS).float())
                cx1 = torch.minimum(bx1, (tx * TS + TS - 1).float())
                cy0 = torch.maximum(by0, (ty * TS).float())
                cy1 = torch.minimum(by1, (ty * TS + TS - 1).float())
                area = ((cx1 - cx0 + 1) * (cy1 - cy0 + 1)).clamp(min=0) * use
                t = (ty * ntx + tx)[use]
                entries[b].index_add_(0, t, torch.ones_like(area[use]))
                cand[b].index_add_(0, t, area[use])
                big = (area > 64)[use]
                cand_big[b].index_add_(0, t[big], area[use][big])
                n_big[b].index_add_(0, t[big], torch.ones_like(area[use][big]))
    return entries.cpu().numpy(), cand.cpu().numpy(), cand_big.cpu().numpy(), n_big.cpu().numpy(), int(ok.sum())


def q(a, ps=(50, 90, 99, 100)):
    return {'p%d' % p: float(np.percentile(a, p)) for p in ps}


def thin_faces(f9, S):
    """faces k_face_setup sends down the band path (face_margin_px > THIN_MARGIN or degenerate), per object"""
    ex0, ey0 = f9[..., 1, 0] - f9[..., 0, 0], f9[..., 1, 1] - f9[..., 0, 1]
    ex1, ey1 = f9[..., 2, 0] - f9[..., 0, 0], f9[..., 2, 1] - f9[..., 0, 1]
    ex2, ey2 = f9[..., 2, 0] - f9[..., 1, 0], f9[..., 2, 1] - f9[..., 1, 1]
    l0, l1, l2 = (ex0 ** 2 + ey0 ** 2).sqrt(), (ex1 ** 2 + ey1 ** 2).sqrt(), (ex2 ** 2 + ey2 ** 2).sqrt()
    lmax = torch.maximum(l0, torch.maximum(l1, l2))
    perim = l0 + l1 + l2
    cmax = f9[..., :2].abs().amax((-1, -2))
    cross = ex0 * ey1 - ex1 * ey0
    area2 = cross.abs() - 9.5367432e-7 * lmax * lmax
    delta = 9.5367432e-7 * (1 + cmax)
    m = 0.004 * max(1.0, S / 1024.0) + 0.5 * S * delta + 0.5 * S * delta * perim * lmax / area2
    thin = (~(area2 > 0)) | (m > 4.0) | ~(m < S)
    front = cross < 0 if False else torch.ones_like(thin)
    return (thin & front).sum(1).tolist()


def main(argv):
    from sdn_hip import ops
    import bench
    if argv and argv[0].endswith('.so'):      # a lab build of the library (tools/build_lab_variant.sh)
        import ctypes
        import sdn_hip
        L = ctypes.CDLL(os.path.abspath(argv[0]))
        sdn_hip._declare(L)
        sdn_hip._lib = L
        print('library:', argv[0])
        argv = argv[1:]
    device = torch.device('cuda:0')
    out = {}
    for mesh in (argv or ['car_like', 'cad_like']):
        f9, scene = frame_faces(device, mesh)
        S = 2 * bench.RENDER_SIZE
        entries, cand, cand_big, n_big, n_ok = stats(f9, S)
        ne = entries[entries > 0]
        # the waves of a tile take equal runs of its list: per-tile cost ~ fixed + batches; the launch ends with its slowest tiles
        rec = {'faces_per_object': int(f9.shape[1]), 'faces_drawn': n_ok, 'tiles_non_empty': int((entries > 0).sum()),
               'tiles': int(entries.size), 'entries_total': float(entries.sum()), 'entries_per_tile': q(ne),
               'candidates_total': float(cand.sum()), 'candidates_per_tile': q(cand[entries > 0]),
               'candidates_large_share': float(cand_big.sum() / max(cand.sum(), 1)),
               'large_entries_total': float(n_big.sum()), 'large_entries_per_tile': q(n_big[entries > 0]),
               'per_object_entries': [float(e.sum()) for e in entries], 'per_object_candidates': [float(c.sum()) for c in cand]}
        rec['thin_faces_per_object_both_windings'] = thin_faces(f9, S)
        bank, cls, params, targets, ptf = scene
        fwd = bench.make_step(device, bank, cls, params, targets, ptf, backward=False)
        import sdn_hip
        sdn_hip.timing_enable(True)
        sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
        with ops.verification(count_work=True):
            with torch.no_grad():
                fwd()
        torch.cuda.synchronize()
        ms, n, _ = sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
        sdn_hip.timing_enable(False)
        rec['counting_launch_us'] = 1e3 * ms / max(n, 1)
        rec['work_counters'] = dict(zip(('candidate_tests', 'passed', 'depth_keys'), ops.last_work() or (0, 0, 0)))
        ck = ops.last_clocks()
        if ck:
            tot = max(ck['total'], 1)
            rec['phase_share'] = {k: round(ck[k] / tot, 4) for k in ('fetch_and_wait', 'lane_boxes', 'wave_boxes', 'thin', 'epilogue')}
            rec['wave_ticks_mean'] = ck['total'] / max(ck['waves'], 1)
            rec['wave_ticks_longest'] = ck['longest_wave']
            rec['waves'] = ck['waves']
        import sdn_hip
        sdn_hip.timing_enable(True)
        sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
        for _ in range(5):
            with torch.no_grad():
                fwd()
        torch.cuda.synchronize()
        ms, n, _ = sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
        sdn_hip.timing_enable(False)
        rec['k_raster_tiles_us'] = 1e3 * ms / max(n, 1)
        out[mesh] = rec
        print(mesh, json.dumps(rec, indent=1))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'tile_stats.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main(sys.argv[1:])
