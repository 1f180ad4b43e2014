#!/bin/bash
# r06d: the 7 x 7 narrow-channel layers on the MFMA head kernel (encoder stem with statistics, head data gradients in 1 / 4 row groups)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_conv_head.py tests/test_gpu_textural.py tests/test_gpu_trainstep.py tests/test_gpu_textural_fullsize.py tests/test_gpu_conv_phases.py tests/test_gpu_dropin.py tests/test_gpu_pipeline_e2e.py -m gpu -q --tb=short -rf -p no:cacheprovider -x > $O/r06d_tests.log 2>&1; echo "tests exit $?" >> $O/r06d_tests.log
tail -5 $O/r06d_tests.log
SDN_WGRAD_STREAM=0 SDN_D_STREAMS=0 timeout 400 python tests/gpu_layer_times.py > $O/r06d_layer_times_serial.log 2>&1
grep -E "^totals|^====|k7" $O/r06d_layer_times_serial.log
timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras > $O/r06d_bench_tex.json 2> $O/r06d_bench_tex.err; cut -c1-200 $O/r06d_bench_tex.json
