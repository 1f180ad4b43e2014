"""Host time of the phases of one textural GAN step on the GPU box (no synchronisation between them: what the host
spends ISSUING each phase), with the per-launch hipEvent timing of bench.py on and off.  Development aid."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import sdn_hip
    dev = torch.device('cuda', 0)
    sys.path.insert(0, os.path.join(ROOT, '3d-sdn_amd', 'textural'))
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    opt = default_options(gpu_ids=[0], batchSize=bench.TEX_BATCH, num_D=3, feat_pose='1', feat_normal='1',
                          no_vgg_loss=True, isTrain=True)
    torch.manual_seed(4321)
    m = Pix2PixHDModel()
    m.initialize(opt)
    label, inst, image, pose, normal = bench.textural_batch(m, dev, 77)

    def step(acc):
        t = [time.perf_counter()]
        losses, _ = m.forward(label, inst.clone(), image, None, pose, normal)
        d = dict(zip(m.loss_names, [x if isinstance(x, int) else torch.mean(x) for x in losses]))
        loss_D = (d['D_fake'] + d['D_real']) * 0.5
        loss_G = d['G_GAN'] + d.get('G_GAN_Feat', 0) + d.get('G_VGG', 0) + d.get('G_L1', 0) + d.get('E_VAE', 0)
        t.append(time.perf_counter())
        m.optimizer_G.zero_grad()
        loss_G.backward()
        t.append(time.perf_counter())
        m.optimizer_G.step()
        t.append(time.perf_counter())
        m.optimizer_D.zero_grad()
        loss_D.backward()
        t.append(time.perf_counter())
        m.optimizer_D.step()
        t.append(time.perf_counter())
        for i in range(5):
            acc[i] += t[i + 1] - t[i]

    for timing in (False, True):
        sdn_hip.timing_enable(timing)
        for _ in range(2):
            step([0] * 5)
        torch.cuda.synchronize()
        acc = [0.0] * 5
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            step(acc)
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        print('hipEvent timing %s: step %.1f ms, host enqueue %.1f ms = forward %.1f + G backward %.1f + G step %.1f + '
              'D backward %.1f + D step %.1f' % (('on ' if timing else 'off'), tot / n * 1e3, enq / n * 1e3,
                                                 *[a / n * 1e3 for a in acc]))
    sdn_hip.timing_enable(False)
    print('cpu count', os.cpu_count(), 'torch threads', torch.get_num_threads())
    os.system("grep -m1 'model name' /proc/cpuinfo; grep -m1 MHz /proc/cpuinfo")


if __name__ == '__main__':
    main()
