"""Development aid (GPU): gradient at every block boundary of the ResNet-18 encoder, HIP path vs fp64 torch modules."""
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


def block64(blk, x):
    idt = x if blk.downsample is None else blk.downsample[1](blk.downsample[0](x))
    out = F.relu(blk.bn1(blk.conv1(x)))
    return F.relu(blk.bn2(blk.conv2(out)) + idt)


size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(3)
net = ResNet18().train()
x = torch.randn(8, 3, size, size)
ref = copy.deepcopy(net).double()
dev = copy.deepcopy(net).cuda()
tr, tg = {}, {}
xr = x.double()
h = F.max_pool2d(F.relu(ref.bn1(ref.conv1(xr))), 3, 2, 1)
tr['stem'] = h
for li in range(1, 5):
    for bi in range(2):
        h = block64(getattr(ref, 'layer%d' % li)[bi], h)
        tr['layer%d.%d' % (li, bi)] = h
pooled = h.mean(dim=(2, 3))
tr['pool'] = pooled
for v in tr.values():
    v.retain_grad()
w = torch.randn_like(pooled)
(pooled * w).sum().backward()
xg = x.cuda()
h = hb.max_pool_3x3_s2(hb.batch_norm(dev.bn1, hb.conv2d(dev.conv1, xg), relu=True))
tg['stem'] = h
for li in range(1, 5):
    for bi in range(2):
        h = getattr(dev, 'layer%d' % li)[bi](h)
        tg['layer%d.%d' % (li, bi)] = h
pooled = hb.global_avg_pool(h)
tg['pool'] = pooled
for v in tg.values():
    v.retain_grad()
(pooled * w.float().cuda()).sum().backward()
for k in tr:
    print('%-9s %-16s value %.1e  grad %.1e' % (k, tuple(tr[k].shape[1:]), rel(tg[k], tr[k]), rel(tg[k].grad, tr[k].grad)))
rp, dp = dict(ref.named_parameters()), dict(dev.named_parameters())
for k in rp:
    if rp[k].grad is not None:
        e = rel(dp[k].grad, rp[k].grad)
        if e > 1e-4:
            print('  param %-28s grad %.1e' % (k, e))
