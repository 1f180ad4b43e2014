#!/usr/bin/env python3
"""Statistics of the silhouette edge gradient's work on the bench frame (development aid, GPU): owners per row, list lengths,
terms -- read back from sdn_rasterize_bwd's workspace (layout of csrc/raster_bwd.hip bwd_layout)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
import bench
import sdn_hip
from sdn_hip import ops

dev = torch.device('cuda:0')
bank, sizes, cls, params, targets, ptf = bench.build_scene(dev, seed=1234)
captured = {}
real = ops.raster_bwd_workspace


def grab(bs, nf, S, device):
    t = real(bs, nf, S, device)
    captured['ws'], captured['dims'] = t, (bs, nf, S)
    return t


ops.raster_bwd_workspace = grab
from derender3d.models import renderer as R
R.Renderer.render_maps = R.Renderer.render_maps_composed      # the path whose Python side owns the workspace
step = bench.make_step(dev, bank, cls, params, targets, ptf, backward=True, pack=False)
step()
torch.cuda.synchronize()
ws, (bs, nf, S) = captured['ws'], captured['dims']
a256 = lambda n: (n + 255) // 256 * 256
n = bs * nf
cap = 4 * n + 65536
off = {0: 0, 1: 256}
off[10] = off[1] + a256(n * 4)
off[2] = off[10] + a256(4 * bs * S * 4)
off[3] = off[2] + a256(n * 4)
off[4] = off[3] + a256(cap * 16)
off[5] = off[4] + a256(cap * 8)
off[6] = off[5] + a256(bs * S * S * 4)
off[7] = off[6] + a256(bs * S * S * 4)
off[8] = off[7] + a256(2 * bs * S * (S + 1) * 2)
off[9] = off[8] + a256(2 * bs * S * S * 2)
off[11] = off[9] + a256(2 * bs * S * S * 4)
nrows = 2 * bs * S
raw = ws.cpu().numpy()
counter = raw[0:4].view(np.uint32)[0]
rc = raw[off[10]:off[10] + nrows * 8].view(np.uint32).reshape(nrows, 2).astype(np.int64)
own = rc.sum(1)
cnt = raw[off[7]:off[7] + nrows * (S + 1) * 2].view(np.uint16).reshape(nrows, S + 1)
total = cnt[:, S].astype(np.int64)
print('chunks', counter, 'edge pixels <=', counter * 8, 'rows', nrows, 'rows with owners', int((own > 0).sum()))
print('owners', int(own.sum()), 'per non-empty row: mean %.1f median %.0f p90 %.0f max %d' % (
    own[own > 0].mean(), np.median(own[own > 0]), np.percentile(own[own > 0], 90), own.max()))
print('list length of rows with owners: mean %.1f median %.0f max %d' % (total[own > 0].mean(), np.median(total[own > 0]), total.max()))
rec = raw[off[11]:off[11] + nrows * 3 * S * 20].view(np.uint32).reshape(nrows, 3 * S, 5)
terms = 0
batch_cost = 0
lens_all = []
for r in np.nonzero(own)[0][::37]:      # a sample of the rows
    for part, m in ((0, rc[r, 0]), (1, rc[r, 1])):
        if not m:
            continue
        kk = rec[r, :m, 3] if part == 0 else rec[r, 3 * S - m:, 3]
        ln = (kk >> 16).astype(np.int64) - (kk & 0xffff).astype(np.int64)
        terms += ln.sum()
        lens_all.append(ln)
        for b in range(0, m, 64):
            batch_cost += ln[b:b + 64].max() * 64
lens = np.concatenate(lens_all)
print('sampled owners %d: range length mean %.1f median %.0f p90 %.0f max %d; lane utilisation of lane=owner batches %.2f'
      % (len(lens), lens.mean(), np.median(lens), np.percentile(lens, 90), lens.max(), terms / max(batch_cost, 1)))
print('estimated terms per frame %.1f M' % (lens.mean() * own.sum() / 1e6))
