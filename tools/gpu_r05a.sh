#!/bin/bash
# r05 first GPU call: the whole GPU suite, the default bench line, and two A/B measurements of this round's kernels
#   (phase launches merged or not: per-record layer times; K1 with / without the hit queue: a short geometric bench).
# usage: tools/gpu_r05a.sh <tag>
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cut -c1-700 $O/${TAG}_bench.json; tail -3 $O/${TAG}_bench.err
SDN_TILE_KERNELS=wfhd timeout 300 python tests/gpu_layer_times.py > $O/${TAG}_layer_times_nophases.log 2>&1
timeout 300 python tests/gpu_layer_times.py > $O/${TAG}_layer_times_phases.log 2>&1
grep -E "^totals|^====" $O/${TAG}_layer_times_nophases.log $O/${TAG}_layer_times_phases.log
SDN_TILE_KERNELS=wfhd timeout 300 python bench.py --no-cpu-baseline --skip-geometric --no-extras --textural-steps 5 > $O/${TAG}_bench_tex_nophases.json 2> $O/${TAG}_bench_tex_nophases.err
python - <<PY
import json
for f in ('${TAG}_bench.json', '${TAG}_bench_tex_nophases.json'):
    try:
        d = json.load(open('$O/' + f))
        print(f, 'value', d.get('value'), 'k1', d.get('value_k1'), 'gan ms', d.get('textural_gan_fwd_bwd_ms'),
              'single', (d.get('roofline_textural') or {}).get('single_stream', {}).get('ms_per_step'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
SDN_K1_PLAIN=1 timeout 300 python bench.py --no-cpu-baseline --skip-textural --steps 20 > $O/${TAG}_bench_k1plain.json 2> $O/${TAG}_bench_k1plain.err
python -c "
import json
d = json.load(open('$O/${TAG}_bench_k1plain.json')); print('K1 plain:', d.get('value_k1'), d.get('k1'))"
