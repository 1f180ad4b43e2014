#!/bin/bash
# r05: k_hlists (second cut) against the r04 pair: geometric parity files, A/B bench, kernel statistics.
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_cad_golden.py tests/test_gpu_k1_coverage.py tests/test_gpu_derender3d.py tests/test_gpu_dropin.py tests/test_gpu_derender_golden.py -m gpu -q --tb=short -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
SDN_EDGE_DENSE_MAPS=1 timeout 300 python bench.py --no-cpu-baseline --skip-textural --no-extras --steps 100 > $O/${TAG}_bench_dense.json 2> $O/${TAG}_bench_dense.err
timeout 300 python bench.py --no-cpu-baseline --skip-textural --no-extras --steps 100 > $O/${TAG}_bench_fused.json 2> $O/${TAG}_bench_fused.err
python - <<PY
import json
for n in ('dense', 'fused'):
    d = json.load(open('$O/${TAG}_bench_%s.json' % n))
    print(n, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'host issue', round(d['host_issue_ms_one_step'], 3))
PY
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o geo -- python $R/bench.py --no-cpu-baseline --skip-textural --no-extras --steps 5 --warmup 2 > $O/${TAG}_prof_geo.log 2>&1
find /tmp/prof_geo -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_geo_kernel_stats.csv \;
grep -E "k_hlists|k_hmap|k_compact" $O/${TAG}_geo_kernel_stats.csv | cut -c1-140
