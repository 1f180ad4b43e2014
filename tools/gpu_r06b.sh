#!/bin/bash
# r06b: textural baseline of the round -- the new batch-4 oracle gates, per-layer times incl. the encoder, the textural bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_textural_fullsize.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "batch4" -s > $O/r06b_tests.log 2>&1; echo "tests exit $?" >> $O/r06b_tests.log
grep -E "passed|failed|batch-4|batch 4|D\(3" $O/r06b_tests.log | head
SDN_WGRAD_STREAM=0 SDN_D_STREAMS=0 timeout 400 python tests/gpu_layer_times.py > $O/r06b_layer_times_serial.log 2>&1
grep -E "^totals|^====" $O/r06b_layer_times_serial.log
timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras > $O/r06b_bench_tex.json 2> $O/r06b_bench_tex.err; cut -c1-400 $O/r06b_bench_tex.json
