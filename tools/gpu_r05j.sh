#!/bin/bash
# r05: sdn_conv_head_mfma second cut (two 32-channel passes, two workgroups per CU): parity, layer times, A/B of the GAN step
TAG=${1:-r05j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv_head.py tests/test_gpu_textural.py tests/test_gpu_trainstep.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee $O/${TAG}_head_tests.log
SDN_WGRAD_STREAM=0 SDN_D_STREAMS=0 timeout 300 python tests/gpu_layer_times.py > $O/${TAG}_layer_times_serial.log 2>&1
grep -E "head mfma|64->3|^totals|^====" $O/${TAG}_layer_times_serial.log
SDN_TILE_KERNELS=wfhdp timeout 400 python bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 5 > $O/${TAG}_bench_tex_nohead.json 2> $O/${TAG}_bench_tex_nohead.err
timeout 400 python bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 5 > $O/${TAG}_bench_tex_head.json 2> $O/${TAG}_bench_tex_head.err
python - <<PY
import json
for n in ('nohead', 'head'):
    d = json.load(open('$O/${TAG}_bench_tex_%s.json' % n))
    r = d['roofline_textural']
    print(n, 'gan', round(d['textural_gan_fwd_bwd_ms'], 2), 'single', round(r['single_stream']['ms_per_step'], 2), 'narrow ms/step', round(r['narrow']['kernel_ms_per_step'], 2), round(r['narrow']['single_stream_kernel_ms_per_step'], 2))
PY
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tex -o tex -- python $R/bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 2 > $O/${TAG}_prof_tex.log 2>&1
find /tmp/prof_tex -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_tex_kernel_stats.csv \;
grep -E "head_mfma|narrow" $O/${TAG}_tex_kernel_stats.csv | cut -c1-150
