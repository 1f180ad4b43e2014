"""Lab: bench.edit_pipeline (configs[4], one GPU) three times in one process; run once per setting of SDN_TILE_KERNELS."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
for k in range(3):
    r = bench.edit_pipeline(dev, 1, 0)
    print('SDN_TILE_KERNELS=%s run %d: %.2f ms per frame (%.3f s) passes %s' % (os.environ.get('SDN_TILE_KERNELS', 'default'), k,
                                                                     r['ms_per_frame_per_gpu'], r['seconds'], ['%.3f' % v for v in r['seconds_of_each_pass']]), flush=True)
