#!/bin/bash
# rocprofv3 kernel stats of the textural leg.  usage: tools/gpu_prof_tex.sh <tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tex -o tex -- python $R/bench.py --no-cpu-baseline --skip-geometric --textural-steps 3 > $O/${TAG}_prof_tex.log 2>&1
find /tmp/prof_tex -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_tex_kernel_stats.csv \;
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/${TAG}_tex_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU ms per step %.1f'%(tot/1e6/4))
for r in rows[:28]:
    print('%-90s %5s calls %9.1f us avg %6.2f %%  %7.2f ms/step'%(r['Name'][:90],r['Calls'],float(r['AverageNs'])/1e3,float(r['Percentage']),float(r['TotalDurationNs'])/1e6/4))
PY
tail -1 $O/${TAG}_prof_tex.log | cut -c1-100
