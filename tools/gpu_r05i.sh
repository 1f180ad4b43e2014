#!/bin/bash
# r05: sdn_conv_head_mfma -- C-ABI parity, the textural suites, per-record layer times and the GAN step with / without it.
TAG=${1:-r05i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv_head.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 | tee $O/${TAG}_head_tests.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
SDN_WGRAD_STREAM=0 SDN_D_STREAMS=0 timeout 300 python tests/gpu_layer_times.py > $O/${TAG}_layer_times_serial.log 2>&1
grep -E "head mfma|narrow|^totals|^====" $O/${TAG}_layer_times_serial.log
SDN_TILE_KERNELS=wfhdp timeout 400 python bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 5 > $O/${TAG}_bench_tex_nohead.json 2> $O/${TAG}_bench_tex_nohead.err
timeout 400 python bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 5 > $O/${TAG}_bench_tex_head.json 2> $O/${TAG}_bench_tex_head.err
python - <<PY
import json
for n in ('nohead', 'head'):
    d = json.load(open('$O/${TAG}_bench_tex_%s.json' % n))
    r = d['roofline_textural']
    print(n, 'gan', round(d['textural_gan_fwd_bwd_ms'], 2), 'single', round(r['single_stream']['ms_per_step'], 2), 'narrow slot', r['narrow'])
PY
