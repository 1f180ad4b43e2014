import sys, os
sys.path.insert(0, '/root/repo')
os.environ['SDN_BENCH_HOST_PROFILE'] = '1'
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
import torch, bench
print(bench.derender3d_loop(torch.device('cuda', 0)))
