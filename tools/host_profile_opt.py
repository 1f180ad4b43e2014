"""Where the HOST time of the test-time optimisation loop goes (bench.derender3d_loop, configs[2]): torch.profiler's CPU table of
one 20-iteration run, on the GPU box.  Development aid (the loop was host-bound at 1.8 ms per iteration before the pose algebra
became one launch each way)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SDN_BENCH_HOST_PROFILE'] = '1'
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
import torch  # noqa: E402

import bench  # noqa: E402

print(bench.derender3d_loop(torch.device('cuda', 0)))
