#!/bin/bash
# r06g: after the packed-FMA fix of k_wgrad_narrow_row: the gates that found it, then the step with the fix / with the r05 asm (lab build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_wgrad_narrow.py tests/test_gpu_conv_head.py tests/test_gpu_trainstep.py tests/test_gpu_textural_fullsize.py tests/test_gpu_textural.py -m gpu -q --tb=short -rf -p no:cacheprovider -s > $O/r06g_tests.log 2>&1; echo "tests exit $?" >> $O/r06g_tests.log
grep -E "passed|failed|FAILED|train step 1|batch-4 backward" $O/r06g_tests.log | cut -c1-220
timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras --textural-steps 8 > $O/r06g_bench_tex.json 2> $O/r06g_bench_tex.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r06g_bench_tex.json'))
print('GAN step ms', d['textural_gan_fwd_bwd_ms'], 'narrow', {k:v for k,v in d['textural'].get('narrow',{}).items() if not isinstance(v,(dict,str))})
P
python tools/lab/head_race.py 192 624 2>&1 | grep -E "HEAD_WIDE|cout|by output"
