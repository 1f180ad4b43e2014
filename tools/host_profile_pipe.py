"""Where the HOST time of the configs[4] pipeline goes (bench.edit_pipeline): torch.profiler's CPU table of one 64-frame pass,
on the GPU box.  Development aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SDN_BENCH_HOST_PROFILE'] = '1'
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
r = bench.edit_pipeline(dev, 1, 0)
print({k: r[k] for k in ('ms_per_frame_per_gpu', 'seconds_of_each_pass')})
