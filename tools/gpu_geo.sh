#!/bin/bash
# Runs on the GPU box (via gpurun): geometric path tests + bench.   usage: tools/gpu_geo.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
tail -5 $O/${TAG}_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cat $O/${TAG}_bench.json
