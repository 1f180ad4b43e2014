"""Lab: which torch operators the frame step (bench.py configs[1]) still issues besides the library's launches, and from where.
usage: python tools/lab/step_ops.py [--mesh cad_like]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mesh', default='cad_like')
    a = ap.parse_args()
    import torch
    from torch.profiler import ProfilerActivity, profile
    import bench
    device = torch.device('cuda', 0)
    bank, sizes, cls, params, targets, ptf = bench.build_scene(device, seed=1234, mesh=a.mesh)
    step = bench.make_step(device, bank, cls, params, targets, ptf, backward=True, pack=False)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    K = 4
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(K):
            step()
        torch.cuda.synchronize()
    print('---- operators per step (device time > 0)')
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
        if e.device_time_total > 0 and e.key.startswith('aten::'):
            print('%-40s calls/step %5.1f  device us/step %7.1f' % (e.key, e.count / K, e.device_time_total / K))
    print('---- by source line (fill / zero / copy / cat / elementwise)')
    for e in sorted(prof.key_averages(group_by_stack_n=6), key=lambda e: -e.device_time_total):
        if e.device_time_total > 0 and e.key.startswith('aten::'):
            where = [s for s in e.stack if 'site-packages' not in s and 'torch/' not in s][:3]
            print('%-28s calls/step %4.1f  us/step %6.1f  %s' % (e.key, e.count / K, e.device_time_total / K, ' <- '.join(where)))


if __name__ == '__main__':
    main()
