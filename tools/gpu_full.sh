#!/bin/bash
# Runs on the GPU box: the whole GPU test suite, the default bench, and rocprofv3 profiles of both legs.
# usage: tools/gpu_full.sh <tag>
TAG=${1:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cat $O/${TAG}_bench.json | cut -c1-1500
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o geo -- python $R/bench.py --no-cpu-baseline --skip-textural --no-extras --steps 5 --warmup 2 > $O/${TAG}_prof_geo.log 2>&1
find /tmp/prof_geo -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_geo_kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tex -o tex -- python $R/bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 3 > $O/${TAG}_prof_tex.log 2>&1
find /tmp/prof_tex -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_tex_kernel_stats.csv \;
head -25 $O/${TAG}_tex_kernel_stats.csv | cut -c1-130
