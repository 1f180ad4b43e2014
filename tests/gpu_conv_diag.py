#!/usr/bin/env python3
"""Diagnostic (development aid): per-stage activation and per-layer gradient errors of the HIP generator against the
fp64 CPU oracle, printed as a table."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
from models import networks as N  # noqa: E402
from oracle import textural_oracle as to  # noqa: E402
from sdn_hip import conv as hc  # noqa: E402


def rl(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    for cfg in os.environ.get('NETS', '48,64,4,9,64,96').split(';'):
        cin, ngf, nd, nb, H, W = [int(v) for v in cfg.split(',')]
        print('==== cin %d ngf %d down %d blocks %d  %dx%d' % (cin, ngf, nd, nb, H, W))
        run(cin, ngf, nd, nb, H, W)


def run(cin, ngf, nd, nb, H, W):
    torch.manual_seed(2)
    G = N.define_G(cin, 3, ngf, 'global', nd, nb)
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    x = torch.randn(1, cin, H, W)
    xo = x.double().clone().requires_grad_(True)
    ps = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if k.endswith('weight') or k.endswith('bias')}
    full = dict(sd)
    full.update(ps)
    acts = []
    yo = to.global_generator(full, xo, nd, nb, collect=acts)
    for a in acts:
        a.retain_grad()
    w = torch.randn(yo.shape, dtype=torch.float64)
    (yo * w).sum().backward()
    G = G.cuda()
    xg = x.cuda().requires_grad_(True)
    yg = G(xg)
    print('output rel L2 %.3e' % rl(yg, yo))
    (yg * w.float().cuda()).sum().backward()
    print('grad input rel L2 %.3e' % rl(xg.grad, xo.grad))
    for k, p in G.named_parameters():
        if k.endswith('weight'):
            print('  grad %-32s %.3e' % (k, rl(p.grad, ps[k].grad)))
    # stage tensors and their gradients, from a manual run of the chain
    chain = G._chain('model', G.model, cin)
    xin = x.cuda().permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        ts, geo = chain.forward(xin, hc.default_precision(), training=False)
    stage_of = [1 + i for i in range(1 + nd)] + [1 + nd + 2 * (b + 1) for b in range(nb)]
    stage_of += [stage_of[-1] + 1 + i for i in range(nd + 1)]
    for a, si in zip(acts, stage_of):
        T = ts[si]
        t = T.data[..., :a.shape[1]].permute(0, 3, 1, 2)
        if T.relu:
            t = torch.relu(t)
        print('  act stage %2d %-18s %.3e' % (si, tuple(a.shape), rl(t, a)))


if __name__ == '__main__':
    main()
