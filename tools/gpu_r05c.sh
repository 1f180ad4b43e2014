#!/bin/bash
# r05 third GPU call: near-to-far tile lists (SDN_RASTER_HIZ=2) against 0 / 1 on the synthetic templates and the real CAD meshes,
# bit-exactness tests under every setting, the pack / unpack microbenchmark, the textural leg after the run-splitting fix.
TAG=${1:-r05c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for H in 0 1 2; do
  SDN_RASTER_HIZ=$H timeout 300 python bench.py --no-cpu-baseline --skip-textural --no-extras --steps 50 > $O/${TAG}_bench_hiz$H.json 2> $O/${TAG}_bench_hiz$H.err
  SDN_RASTER_HIZ=$H timeout 300 python tools/raster_hiz_lab.py 2>&1 | tail -1 | tee -a $O/${TAG}_hiz_real_meshes.log
done
SDN_RASTER_HIZ=2 timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_cad_golden.py tests/test_gpu_derender3d.py tests/test_gpu_dropin.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -5 | tee -a $O/${TAG}_hiz_tests.log
python - <<PY
import json
for h in (0, 1, 2):
    try:
        d = json.load(open('$O/${TAG}_bench_hiz%d.json' % h))
        print('HIZ', h, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'k_raster_tiles us', round(d['roofline_raster_fwd']['avg_launch_us'], 1),
              'cand', d.get('roofline_alu', {}).get('candidate_pixel_tests'), 'passed', d.get('roofline_alu', {}).get('tests_passed'), 'keys', d.get('roofline_alu', {}).get('depth_keys'))
    except Exception as e:
        print('HIZ', h, 'unreadable', e)
PY
timeout 300 python tools/pack_lab.py 2>&1 | grep -v Warning | tee $O/${TAG}_pack_lab.log
timeout 600 python bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 5 > $O/${TAG}_bench_tex.json 2> $O/${TAG}_bench_tex.err
python -c "
import json
d = json.load(open('$O/${TAG}_bench_tex.json')); print('tex: gan', d.get('textural_gan_fwd_bwd_ms'), 'single', d['roofline_textural']['single_stream']['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tex -o tex -- python $R/bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 2 > $O/${TAG}_prof_tex.log 2>&1
find /tmp/prof_tex -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_tex_kernel_stats.csv \;
grep -E "k_weights_multi|k_unpack|k_pack" $O/${TAG}_tex_kernel_stats.csv
