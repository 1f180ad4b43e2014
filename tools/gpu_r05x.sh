#!/bin/bash
# r05x: every N > 1 code path of the default bench line (geometric leg with the map exchange, textural leg, configs[4] with its
# all_gather) executed by TWO ranks on one GPU over gloo (--share-gpu: a development run, not a measurement)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --textural-steps 1 > $O/r05x_share2_full.json 2> $O/r05x_share2_full.err
echo "rc $?"; tail -3 $O/r05x_share2_full.err | cut -c1-300
python - <<PY
import json
d = json.load(open('$O/r05x_share2_full.json'))
print('ranks_seen', d.get('ranks_seen'), 'value', d.get('value'), 'gan ms', d.get('textural_gan_fwd_bwd_ms'))
print('exchange', d.get('exchange'))
e = d.get('edit_pipeline', {})
print('edit_pipeline', {k: e.get(k) for k in ('ms_per_frame_per_gpu', 'frames_per_s', 'allgather_payload_bytes_per_rank', 'gathered_maps_checksum', 'seconds_of_each_pass', 'error')})
PY
