#!/bin/bash
# r06e: generator update (Adam + eager re-pack) on a side stream beside the discriminator's backward pass: gates, then A/B of the step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_conv_head.py tests/test_gpu_trainstep.py tests/test_gpu_textural.py tests/test_gpu_dropin.py tests/test_gpu_pipeline_e2e.py tests/test_gpu_textural_fullsize.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/r06e_tests.log 2>&1; echo "tests exit $?" >> $O/r06e_tests.log
tail -5 $O/r06e_tests.log
for U in 1 0 1 0; do
  SDN_UPDATE_STREAM=$U timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras --textural-steps 8 > $O/r06e_bench_tex_u$U.json 2> $O/r06e_bench_tex_u$U.err; echo "UPDATE_STREAM=$U $(cut -c1-60 $O/r06e_bench_tex_u$U.json)"
done
