#!/usr/bin/env python3
"""Development aid: per-layer kernel times of one generator / discriminator forward+backward at the BASELINE size."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
from models import networks as N  # noqa: E402
from sdn_hip import conv as hc  # noqa: E402


def run(name, net, x, loss):
    for _ in range(2):
        xx = x.clone().requires_grad_(True)
        loss(net(xx)).backward()
    hc.PROFILE = []
    xx = x.clone().requires_grad_(True)
    loss(net(xx)).backward()
    prof, hc.PROFILE = hc.PROFILE, None
    print('==== %s' % name)
    tot = {}
    for what, desc, ms, fl in prof:
        tot[what] = tot.get(what, 0.0) + ms
        print('%-9s %-44s %8.3f ms %s' % (what, desc, ms, ('%7.1f TFLOP/s' % (fl / ms / 1e9)) if fl else ''))
    print('totals: ' + '  '.join('%s %.2f ms' % kv for kv in tot.items()))


def main():
    bs = int(os.environ.get('BS', '4'))
    torch.manual_seed(0)
    G = N.define_G(48, 3, 64, 'global', 4, 9).cuda()
    D = N.define_D(18, 64, 3, 'instance', False, 3, True).cuda()
    run('G', G, torch.randn(bs, 48, 384, 1248, device='cuda'), lambda y: y.sum())
    run('D', D, torch.randn(bs, 18, 384, 1248, device='cuda'), lambda r: sum(f.mean() for s in r for f in s))
    # the feature encoder (networks.py:286-346) with ten rectangular instances per image, as Pix2PixHDModel calls it
    E = N.define_G(3, 5, 16, 'encoder', 4).cuda()
    inst = torch.zeros(bs, 1, 384, 1248, device='cuda')
    for k in range(10):
        inst[:, :, 30 * k:30 * k + 60, 100 * k:100 * k + 200] = 1000 * (k + 1)
    run('E', lambda x: E(x, inst), torch.randn(bs, 3, 384, 1248, device='cuda'), lambda y: y.sum())


if __name__ == '__main__':
    main()
