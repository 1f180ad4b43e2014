"""configs[2] numbers (bench.derender3d_loop) with the weight-gradient side stream on and off, same box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda', 0)
for rep in range(2):
    for mode in ('0', '1'):
        os.environ['SDN_WGRAD_STREAM'] = mode
        r = bench.derender3d_loop(dev, n_opts=20)
        print('SDN_WGRAD_STREAM=%s: encoder fwd %.2f ms, inference %.2f, optimisation %.1f, train step %.2f ms'
              % (mode, r['encoder_fwd_ms'], r['inference_ms'], r['optimisation_ms'], r['train_step_ms']), flush=True)
