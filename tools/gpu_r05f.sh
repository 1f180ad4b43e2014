#!/bin/bash
# r05: full suite, then the fused list kernel (k_hlists) against the r04 pair (SDN_EDGE_DENSE_MAPS=1), then the default bench.
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
SDN_EDGE_DENSE_MAPS=1 timeout 300 python bench.py --no-cpu-baseline --skip-textural --no-extras --steps 100 > $O/${TAG}_bench_dense.json 2> $O/${TAG}_bench_dense.err
timeout 300 python bench.py --no-cpu-baseline --skip-textural --no-extras --steps 100 > $O/${TAG}_bench_fused.json 2> $O/${TAG}_bench_fused.err
SDN_EDGE_DENSE_MAPS=1 timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
python - <<PY
import json
for n in ('dense', 'fused'):
    d = json.load(open('$O/${TAG}_bench_%s.json' % n))
    print(n, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'host issue', round(d['host_issue_ms_one_step'], 3))
PY
timeout 900 python bench.py --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -c "
import json
d = json.load(open('$O/${TAG}_bench.json')); print('bench: value', d['value'], 'k1', d.get('value_k1'), 'car', d.get('value_car_like'), 'gan', d.get('textural_gan_fwd_bwd_ms'), 'single', d['roofline_textural']['single_stream']['ms_per_step'], 'loop', d['derender3d_loop']['optimisation_objects_per_s'])"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o geo -- python $R/bench.py --no-cpu-baseline --skip-textural --no-extras --steps 5 --warmup 2 > $O/${TAG}_prof_geo.log 2>&1
find /tmp/prof_geo -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_geo_kernel_stats.csv \;
head -14 $O/${TAG}_geo_kernel_stats.csv | cut -c1-100
