#!/bin/bash
# r05 fourth GPU call: full suite on the current tree, the default bench, the N = 2 code paths on ONE GPU (--share-gpu over gloo:
# a development run, never a measurement) in both exchange modes and with the frame payload, the pack lab.
TAG=${1:-r05d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -30
timeout 300 python tools/pack_lab.py 2>&1 | grep -v Warning | tee $O/${TAG}_pack_lab.log
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python -c "
import json
d = json.load(open('$O/${TAG}_bench.json')); print('bench: value', d['value'], 'k1', d.get('value_k1'), 'car', d.get('value_car_like'), 'gan', d.get('textural_gan_fwd_bwd_ms'), 'single', d['roofline_textural']['single_stream']['ms_per_step'], 'edit', d.get('edit_pipeline', {}).get('ms_per_frame_per_gpu'), d.get('edit_pipeline', {}).get('gathered_maps_checksum'), d.get('edit_pipeline', {}).get('error'))"
for MODE in all_gather p2p; do
  SDN_EXCHANGE=$MODE timeout 600 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --skip-textural --no-extras > $O/${TAG}_share2_$MODE.json 2> $O/${TAG}_share2_$MODE.err
  echo "share-gpu world 2, $MODE: rc $?"; python -c "
import json
d = json.load(open('$O/${TAG}_share2_$MODE.json')); print(d.get('ranks_seen'), d.get('value'), d.get('exchange'))" 2>&1 | tail -2; tail -3 $O/${TAG}_share2_$MODE.err
done
SDN_EXCHANGE_PAYLOAD=frame timeout 600 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --skip-textural --no-extras > $O/${TAG}_share2_frame.json 2> $O/${TAG}_share2_frame.err
echo "share-gpu world 2, frame payload: rc $?"; python -c "
import json
d = json.load(open('$O/${TAG}_share2_frame.json')); print(d.get('ranks_seen'), d.get('value'), d.get('exchange'))" 2>&1 | tail -2; tail -3 $O/${TAG}_share2_frame.err
