"""Lab: the generator head (ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh) alone, forward + backward at 384 x 1248, batch 4: weight and
input gradients against float64 for SDN_HEAD_WIDE x SDN_WGRAD_STREAM.  (r06f: model.38.weight off by 1e-3 only with both on.)"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')]
from sdn_hip import conv as hc  # noqa: E402


def run(wide, wstream, seed=0, H=384, W=1248, N=4, extra=1, CO=3):
    os.environ['SDN_HEAD_WIDE'] = wide
    os.environ['SDN_WGRAD_STREAM'] = wstream
    torch.manual_seed(seed)
    # a 3x3 conv in front so that the chain has >= 2 stages (the weight-gradient side stream is only used then)
    mods = [nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.ReflectionPad2d(3), nn.Conv2d(64, CO, 7), nn.Tanh()]
    for m in mods:
        m.cuda()
    stages, last = hc.compile_sequential(mods)
    chain = hc.ConvChain(stages, [last], 64)
    x = torch.randn(N, 64, H, W, device='cuda', requires_grad=True)
    w = torch.randn(N, CO, H, W, device='cuda')
    y = chain(x)[0]
    (y * w).sum().backward()
    torch.cuda.synchronize()
    gw = mods[3].weight.grad.double().cpu()
    gx = x.grad.double().cpu()
    # float64 reference on the CPU for the head only: needs the hidden activation; recompute with torch on CPU
    xd = x.detach().double().cpu().requires_grad_(True)
    c0w, c0b = mods[0].weight.detach().double().cpu(), mods[0].bias.detach().double().cpu()
    hw = mods[3].weight.detach().double().cpu().requires_grad_(True)
    hb = mods[3].bias.detach().double().cpu()
    h = F.relu(F.conv2d(xd, c0w, c0b, padding=1))
    yr = torch.tanh(F.conv2d(F.pad(h, (3, 3, 3, 3), mode='reflect'), hw, hb))
    (yr * w.double().cpu()).sum().backward()
    e_w = float((gw - hw.grad).norm() / hw.grad.norm())
    e_x = float((gx - xd.grad).norm() / xd.grad.norm())
    print('HEAD_WIDE=%s WGRAD_STREAM=%s: head weight gradient rel %.2e, input gradient rel %.2e' % (wide, wstream, e_w, e_x), flush=True)
    if e_w > 5e-5:
        d = (gw - hw.grad)                       # [3, 64, 7, 7]
        ref = hw.grad
        print('   by output row r   :', ['%.1e' % float(d[r].norm() / ref[r].norm()) for r in range(CO)])
        print('   by channel chunk  :', ['%.1e' % float(d[:, 16 * c:16 * c + 16].norm() / ref[:, 16 * c:16 * c + 16].norm()) for c in range(4)])
        print('   by tap row dy     :', ['%.1e' % float(d[:, :, k].norm() / ref[:, :, k].norm()) for k in range(7)])
        print('   by tap column dx  :', ['%.1e' % float(d[:, :, :, k].norm() / ref[:, :, :, k].norm()) for k in range(7)])
        rel = (d.abs() / ref.abs().clamp(min=1e-12)).flatten()
        print('   elements off by > 1e-4 relative: %d of %d; sign of d.ref: %.3f' % (int((rel > 1e-4).sum()), rel.numel(),
              float((d * ref).sum() / (d.norm() * ref.norm()))))


if __name__ == '__main__':
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (192, 624)
    print('SDN_DEBUG_FORK =', os.environ.get('SDN_DEBUG_FORK'), ' SDN_DEBUG_SYNC_CODES =', os.environ.get('SDN_DEBUG_SYNC_CODES'))
    for co in (3, 4, 6, 8):
        print('cout', co)
        run('1', '1', H=H, W=W, CO=co)
