"""Where the HOST time of one geometric frame step goes (bench.make_step), on the GPU box: torch.profiler's CPU-side table of
a few steps issued into an empty queue (synchronised before each step, so the numbers are issue costs, not queue waits).
Development aid for the issue-bound frame step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    bank, sizes, cls, params, targets, ptf = bench.build_scene(dev, 1234)
    step = bench.make_step(dev, bank, cls, params, targets, ptf, pack=False)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # plain timing: forward / backward split
    fwd = bench.make_step(dev, bank, cls, params, targets, ptf, backward=False, pack=False)
    t_f, t_s = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            fwd()
        t_f.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t_s.append(time.perf_counter() - t0)
    print('host issue: forward only (no_grad) %.3f ms, forward + backward %.3f ms (medians of 20)'
          % (1e3 * sorted(t_f)[10], 1e3 * sorted(t_s)[10]))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(10):
            torch.cuda.synchronize()
            step()
    torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=60))


if __name__ == '__main__':
    main()
