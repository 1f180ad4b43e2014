"""Diagnostic: (1) torch's avg_pool2d backward on ROCm for a channels-last-strided view; (2) the discriminator's image gradient
when the image part is such a view (what Pix2PixHDModel hands it: the generator's output)."""
import os, sys, json, tempfile
import numpy as np, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    sys.path.insert(0, p)
os.environ.setdefault('SDN_DETERMINISTIC', '1'); os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
torch.manual_seed(0)
N, H, W = 2, 32, 48
ds = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)
w1 = torch.randn(N, 3, H, W).cuda(); w2 = torch.randn(N, 3, H // 2, W // 2).cuda()
for C in (16, 3):
    base = torch.randn(N, H, W, C).cuda().requires_grad_(True)
    fake = base[..., :3].permute(0, 3, 1, 2)
    ((fake * w1).sum() + (ds(fake) * w2).sum()).backward()
    g1 = base.grad[..., :3].permute(0, 3, 1, 2).clone()
    f2 = fake.detach().contiguous().clone().requires_grad_(True)
    ((f2 * w1).sum() + (ds(f2) * w2).sum()).backward()
    print('torch avg_pool2d backward, view of [N,H,W,%d]: max diff vs contiguous %.3e (max |g| %.3e)' % (C, float((g1 - f2.grad).abs().max()), float(f2.grad.abs().max())))
    y1 = ds(fake); y2 = ds(f2)
    print('   forward diff %.3e  strides in %s out %s' % (float((y1 - y2).abs().max()), tuple(fake.stride()), tuple(y1.stride())))
from test_gpu_trainstep import GOLD, _model
z = np.load(GOLD)
m, _ = _model(z, tempfile.mkdtemp())
D = m.netD
lab = torch.randn(N, 6, H, W).cuda()
base = torch.randn(N, H, W, 16).cuda()
def gan(res): return sum(((s[-1] - 1.0) ** 2).mean() for s in res)
img_c = base[..., :3].permute(0, 3, 1, 2).contiguous().clone().requires_grad_(True)
gan(D([lab, img_c], detach_weights=True)).backward()
b2 = base.clone().requires_grad_(True)
img_v = b2[..., :3].permute(0, 3, 1, 2)
img_v.retain_grad()
gan(D([lab, img_v], detach_weights=True)).backward()
print('D image gradient, strided view vs contiguous: rel %.3e' % float((img_v.grad - img_c.grad).norm() / img_c.grad.norm()))
# one scale only (no pooling)
D1 = m.netD
img_c = base[..., :3].permute(0, 3, 1, 2).contiguous().clone().requires_grad_(True)
r = D1([lab, img_c], detach_weights=True)
((r[0][-1] - 1) ** 2).mean().backward()
b3 = base.clone().requires_grad_(True)
img_v = b3[..., :3].permute(0, 3, 1, 2); img_v.retain_grad()
r = D1([lab, img_v], detach_weights=True)
((r[0][-1] - 1) ** 2).mean().backward()
print('  finest scale only: rel %.3e' % float((img_v.grad - img_c.grad).norm() / img_c.grad.norm()))
img_c = base[..., :3].permute(0, 3, 1, 2).contiguous().clone().requires_grad_(True)
r = D1([lab, img_c], detach_weights=True)
((r[1][-1] - 1) ** 2).mean().backward()
b4 = base.clone().requires_grad_(True)
img_v = b4[..., :3].permute(0, 3, 1, 2); img_v.retain_grad()
r = D1([lab, img_v], detach_weights=True)
((r[1][-1] - 1) ** 2).mean().backward()
print('  pooled scale only: rel %.3e' % float((img_v.grad - img_c.grad).norm() / img_c.grad.norm()))
