"""Lab (a measurement, not a product path): what the one host synchronisation of a GAN train step costs -- torch.unique in the encoder's
instance pooling (networks.py:317, the reference goes through numpy on the host) returns a tensor of data-dependent size.  The step is
timed as the bench times it, then again with torch.unique's result REPLAYED from the first call (valid only because the lab feeds the
same instance map every step): the difference is the price of the drain + the unique kernels."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')]
import bench  # noqa: E402
from models.pix2pixHD_model import Pix2PixHDModel, default_options  # noqa: E402

device = torch.device('cuda', 0)
opt = default_options(gpu_ids=[0], batchSize=bench.TEX_BATCH, num_D=3, feat_pose='1', feat_normal='1', no_vgg_loss=True, isTrain=True)
torch.manual_seed(4321)
model = Pix2PixHDModel()
model.initialize(opt)
label, inst, image, pose, normal = bench.textural_batch(model, device, 77)


def step():
    return model.train_step(label, inst.clone(), image, None, pose, normal)


def timed(k=8):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


real_unique = torch.unique
for rnd in range(2):
    print('with torch.unique (one host sync per step): %.2f ms' % timed(), flush=True)
    kept = {}

    def replay(x, *a, **k):
        if 'r' not in kept:
            kept['r'] = real_unique(x, *a, **k)
        return kept['r']
    torch.unique = replay
    print('unique replayed from the first call (no sync): %.2f ms' % timed(), flush=True)
    torch.unique = real_unique
