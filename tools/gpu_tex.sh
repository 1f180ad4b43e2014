#!/bin/bash
# Runs on the GPU box: textural parity tests (all failures reported) + conv micro-benchmark.   usage: tools/gpu_tex.sh <tag>
TAG=${1:-tex}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_textural.py -m gpu -q --tb=short -rf -s > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error|rel L2" $O/${TAG}_tests.log | head -60
timeout 600 python tests/gpu_conv_bench.py > $O/${TAG}_convbench.log 2>&1; tail -8 $O/${TAG}_convbench.log
