"""A/B of the discriminator's side streams (SDN_D_STREAMS) on the full GAN step, same process, same box."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    sys.path.insert(0, os.path.join(ROOT, '3d-sdn_amd', 'textural'))
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    opt = default_options(gpu_ids=[0], batchSize=bench.TEX_BATCH, num_D=3, feat_pose='1', feat_normal='1',
                          no_vgg_loss=True, isTrain=True)
    torch.manual_seed(4321)
    m = Pix2PixHDModel()
    m.initialize(opt)
    label, inst, image, pose, normal = bench.textural_batch(m, dev, 77)

    def step():
        return m.train_step(label, inst.clone(), image, None, pose, normal)
    res = {}
    var = 'SDN_WGRAD_STREAM' if '--wgrad' in sys.argv else 'SDN_D_STREAMS'
    print('switching %s' % var)
    for rep in range(3):
        for mode in ('0', '1'):
            os.environ[var] = mode
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = step()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append((time.perf_counter() - t0) / 5 * 1e3)
    print('GAN step ms, one stream : %s' % ['%.2f' % v for v in res['0']])
    print('GAN step ms, side streams: %s' % ['%.2f' % v for v in res['1']])
    print({k: float(v) for k, v in out.items()})


def check():
    """SDN_DETERMINISTIC=1: two train steps from the same initial state must leave bit-identical parameters whether the
    coarse discriminator columns run on side streams or not (a missing stream dependency would show up here)."""
    os.environ['SDN_DETERMINISTIC'] = '1'
    dev = torch.device('cuda', 0)
    sys.path.insert(0, os.path.join(ROOT, '3d-sdn_amd', 'textural'))
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    sums = {}
    for mode in ('0', '0', '1', '1'):
        os.environ['SDN_D_STREAMS'] = mode
        opt = default_options(gpu_ids=[0], batchSize=bench.TEX_BATCH, num_D=3, feat_pose='1', feat_normal='1',
                              no_vgg_loss=True, isTrain=True)
        torch.manual_seed(4321)
        m = Pix2PixHDModel()
        m.initialize(opt)
        label, inst, image, pose, normal = bench.textural_batch(m, dev, 77)
        for _ in range(2):
            out = m.train_step(label, inst.clone(), image, None, pose, normal)
        torch.cuda.synchronize()
        cs = [float(p.detach().double().sum()) for net in (m.netG, m.netD, m.netE) for p in net.parameters()]
        rs = [float(b.double().sum()) for b in m.netD.buffers()]
        sums.setdefault(mode, []).append((cs, rs, {k: float(v) for k, v in out.items()}))
        del m
        torch.cuda.empty_cache()
    import numpy as np

    def dist(u, v):   # relative L2 distance of the parameter-checksum vectors / of the losses
        u0, v0 = np.asarray(u[0]), np.asarray(v[0])
        lu, lv = np.asarray(list(u[2].values())), np.asarray(list(v[2].values()))
        return float(np.linalg.norm(u0 - v0) / np.linalg.norm(u0)), float(np.abs(lu - lv).max())
    a0, a1, b0, b1 = sums['0'][0], sums['0'][1], sums['1'][0], sums['1'][1]
    print('one stream vs one stream  : checksums %.3e, losses %.3e' % dist(a0, a1))
    print('side streams vs side      : checksums %.3e, losses %.3e' % dist(b0, b1))
    print('one stream vs side streams: checksums %.3e, losses %.3e' % dist(a0, b0))
    print('one stream vs side (2)    : checksums %.3e, losses %.3e' % dist(a1, b1))


if __name__ == '__main__':
    check() if '--check' in sys.argv else main()
