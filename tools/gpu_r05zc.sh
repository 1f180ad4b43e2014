#!/bin/bash
# GPU busy time of one configs[4] pass: rocprofv3 kernel statistics of one edit_pipeline call (1 warm-up + 3 timed passes = 4 passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cat > /tmp/one_pipe.py <<PY
import sys, torch
sys.path.insert(0, '$R')
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
r = bench.edit_pipeline(dev, 1, 0)
print('PIPE', r['ms_per_frame_per_gpu'], r['seconds_of_each_pass'])
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe -o pipe -- python /tmp/one_pipe.py > $O/r05zc_prof_pipe.log 2>&1
grep PIPE $O/r05zc_prof_pipe.log
find /tmp/prof_pipe -name '*kernel_stats.csv' -exec cp {} $O/r05zc_pipe_kernel_stats.csv \;
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/r05zc_pipe_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time summed over 4 passes: %.1f ms -> %.1f ms per pass = %.2f ms per frame; launches per frame %.0f' % (tot / 1e6, tot / 4e6, tot / 4e6 / 64, sum(int(r['Calls']) for r in rows) / 4 / 64))
for r in rows[:22]:
    print('%-70s %6d calls %8.1f ms  %5.1f %%' % (r['Name'][:70], int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
PY
