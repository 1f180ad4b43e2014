#!/bin/bash
# r05 second GPU call: the fixture-dependent tests + this call's new tests, the hierarchical depth cull A/B (synthetic templates
# through bench.py, the real CAD meshes through tools/raster_hiz_lab.py), the edit pipeline, kernel statistics of both legs.
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_cad_golden.py tests/test_gpu_k1_coverage.py tests/test_gpu_composite.py tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_derender3d.py tests/test_gpu_pipeline_e2e.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
for H in 0 1 2; do
  SDN_RASTER_HIZ=$H timeout 300 python bench.py --no-cpu-baseline --skip-textural --no-extras --steps 50 > $O/${TAG}_bench_hiz$H.json 2> $O/${TAG}_bench_hiz$H.err
  SDN_RASTER_HIZ=$H timeout 300 python tools/raster_hiz_lab.py 2>&1 | tail -1 | tee -a $O/${TAG}_hiz_real_meshes.log
  # every raster / renderer parity test under this setting too (bit-exactness must not depend on the switch)
  SDN_RASTER_HIZ=$H timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_cad_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1 | tee -a $O/${TAG}_hiz_tests.log
done
python - <<PY
import json
for h in (0, 1, 2):
    try:
        d = json.load(open('$O/${TAG}_bench_hiz%d.json' % h))
        print('HIZ', h, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'k_raster_tiles us', round(d['roofline_raster_fwd']['avg_launch_us'], 1),
              'cand', d.get('roofline_alu', {}).get('candidate_pixel_tests'), 'keys', d.get('roofline_alu', {}).get('depth_keys'))
    except Exception as e:
        print('HIZ', h, 'unreadable', e)
PY
SDN_RASTER_HIZ=1 timeout 300 python bench.py --no-cpu-baseline --skip-textural --steps 20 > $O/${TAG}_bench_geo_full.json 2> $O/${TAG}_bench_geo_full.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -c "
import json
d = json.load(open('$O/${TAG}_bench.json')); print('full bench: value', d['value'], 'k1', d.get('value_k1'), 'car', d.get('value_car_like'), 'gan', d.get('textural_gan_fwd_bwd_ms'), 'edit', d.get('edit_pipeline'))"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tex -o tex -- python $R/bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 2 > $O/${TAG}_prof_tex.log 2>&1
find /tmp/prof_tex -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_tex_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o geo -- python $R/bench.py --no-cpu-baseline --skip-textural --no-extras --steps 5 --warmup 2 > $O/${TAG}_prof_geo.log 2>&1
find /tmp/prof_geo -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_geo_kernel_stats.csv \;
head -12 $O/${TAG}_geo_kernel_stats.csv | cut -c1-110
