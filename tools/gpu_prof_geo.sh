#!/bin/bash
# Per-step kernel table of the frame step: rocprofv3 kernel stats of tools/prof_geo.py (K identical steps), totals / K.
#   usage: tools/gpu_prof_geo.sh <tag> [steps] [mesh]
TAG=$1; K=${2:-20}; MESH=${3:-car_like}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_geo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o geo -- python $R/tools/prof_geo.py --steps $K --mesh $MESH > $O/${TAG}_prof_geo.log 2>&1
find /tmp/prof_geo -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_geo_kernel_stats.csv \;
grep PROF_GEO $O/${TAG}_prof_geo.log
python - <<PY | tee $O/${TAG}_geo_step_table.md
import csv
rows=list(csv.DictReader(open('$O/${TAG}_geo_kernel_stats.csv')))
K=$K
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step (sum over kernels): %.1f us; %.1f launches per step' % (tot/1e3/K, sum(int(r['Calls']) for r in rows)/K))
print()
print('| kernel | launches / step | us / launch | us / step |')
print('|---|---|---|---|')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print('| \`%s\` | %.2f | %.1f | %.1f |' % (r['Name'][:80], int(r['Calls'])/K, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/K))
PY
