#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes of `bench.py <args>`.
# usage: tools/gpu_prof.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $O/${TAG}_prof.log 2>&1
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $O/${TAG}_pmc_$C.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_${TAG}_$C sdn:: $TAG > $O/${TAG}_pmc_$C.json
done
head -12 $O/${TAG}_kernel_stats.csv | cut -c1-150
cat $O/${TAG}_pmc_FETCH_SIZE.json | head -40
