"""Development aid (GPU): gradients at every intermediate tensor of a stride-2 BasicBlock, HIP path vs fp64."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
    sys.path.insert(0, p)
os.environ['SDN_ALLOW_RANDOM_INIT'] = '1'
import torch
import torch.nn.functional as F
from sdn_hip import bnnet as hb
from derender3d.models.resnet import ResNet18


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


torch.manual_seed(0)
net = ResNet18().train()
for name, shape in (('layer4.1', (8, 2, 2)), ('layer4.1', (8, 6, 7)), ('layer4.1', (1, 6, 7)), ('layer3.1', (8, 4, 4)), ('layer1.0', (8, 2, 2))):
    li, bi = name.split('.')
    blk = getattr(net, li)[int(bi)]
    cin = blk.conv1.in_channels
    x = torch.relu(torch.randn(shape[0], cin, shape[1], shape[2]))
    ref = copy.deepcopy(blk).double()
    dev = copy.deepcopy(blk).cuda()
    T = {}

    def run(b, x, conv, bn, tag):
        t = {}
        t['x'] = x
        if b.downsample is not None:
            t['dsc'] = conv(b.downsample[0], x)
            t['idt'] = bn(b.downsample[1], t['dsc'], None, False)
        else:
            t['idt'] = x
        t['c1'] = conv(b.conv1, x)
        t['b1'] = bn(b.bn1, t['c1'], None, True)
        t['c2'] = conv(b.conv2, t['b1'])
        t['out'] = bn(b.bn2, t['c2'], t['idt'], True)
        for k, v in t.items():
            if v.requires_grad and not v.is_leaf:
                v.retain_grad()
        return t
    xr = x.double().requires_grad_(True)
    tr = run(ref, xr, lambda m, v: m(v), lambda m, v, r, relu: (F.relu(m(v) + r) if relu else m(v) + r) if r is not None else (F.relu(m(v)) if relu else m(v)), 'ref')
    gy = torch.randn_like(tr['out'])
    tr['out'].backward(gy)
    xg = x.cuda().requires_grad_(True)
    tg = run(dev, xg, hb.conv2d, lambda m, v, r, relu: hb.batch_norm(m, v, res=r, relu=relu), 'hip')
    tg['out'].backward(gy.float().cuda())
    print(name, shape)
    for k in tr:
        g1 = tg[k].grad if k != 'x' else xg.grad
        g0 = tr[k].grad if k != 'x' else xr.grad
        print('  %-4s value %.1e  grad %.1e' % (k, rel(tg[k], tr[k]), rel(g1, g0) if g1 is not None and g0 is not None else -1))
    rp, dp = dict(ref.named_parameters()), dict(dev.named_parameters())
    for k in rp:
        print('  param %-22s grad %.1e' % (k, rel(dp[k].grad, rp[k].grad)))
