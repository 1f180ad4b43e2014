#!/bin/bash
# PMC passes (L2 hit/miss, fabric fetch/write) of the conv kernels in the textural leg.  usage: tools/gpu_pmc_tex.sh <tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$N -o t -- python $R/bench.py --no-cpu-baseline --skip-geometric --textural-steps 1 > $O/${TAG}_pmc_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_$N k_conv > $O/${TAG}_pmc_tex_$N.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/${TAG}_pmc_tex_*.json')):
    d=json.load(open(f))
    for k,v in d.items(): print(k[:60], {c:(round(x['mean']),x['dispatches']) for c,x in v.items()})
PY
