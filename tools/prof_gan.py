"""One process = W + K identical GAN train steps (bench.py's textural configuration), nothing else: the target of
`rocprofv3 --kernel-trace --stats` when a per-step kernel table is wanted (tools/gpu_prof_gan.sh divides every total by the
step count this script prints -- all steps launch the same kernels).

    python tools/prof_gan.py [--steps 4] [--single-stream]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')]
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--single-stream', action='store_true')
    a = ap.parse_args()
    if a.single_stream:
        os.environ['SDN_D_STREAMS'] = os.environ['SDN_WGRAD_STREAM'] = '0'
    import torch
    import bench
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    device = torch.device('cuda', 0)
    opt = default_options(gpu_ids=[0], batchSize=bench.TEX_BATCH, num_D=3, feat_pose='1', feat_normal='1', no_vgg_loss=True,
                          isTrain=True)
    torch.manual_seed(4321)
    model = Pix2PixHDModel()
    model.initialize(opt)
    label, inst, image, pose, normal = bench.textural_batch(model, device, 77)
    model.train_step(label, inst.clone(), image, None, pose, normal)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps - 1):
        model.train_step(label, inst.clone(), image, None, pose, normal)
    torch.cuda.synchronize()
    print('PROF_GAN steps %d  ms_per_step %.2f  single_stream %d' % (a.steps, (time.perf_counter() - t0) / max(a.steps - 1, 1) * 1e3,
                                                                     int(a.single_stream)))


if __name__ == '__main__':
    main()
