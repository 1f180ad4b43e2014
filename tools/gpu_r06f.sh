#!/bin/bash
# r06f: which change moved the head-layer gradients?  (trainstep E/model.17.bias 2.25e-4, fullsize default model.38.weight 9.9e-4)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for V in "SDN_HEAD_WIDE=1" "SDN_HEAD_WIDE=0" "SDN_HEAD_WIDE=1 SDN_WGRAD_STREAM=0" ; do
  echo "==== $V"
  env $V timeout 600 python -m pytest tests/test_gpu_textural_fullsize.py -m gpu -q --tb=line -p no:cacheprovider -k "batch4_backward" -s 2>&1 | grep -E "batch-4 backward|passed|failed"
  env $V timeout 600 python -m pytest tests/test_gpu_trainstep.py -m gpu -q --tb=line -p no:cacheprovider -k "reproduces" -s 2>&1 | grep -E "train step 1|step 1 E/model.17|passed|failed"
done > $O/r06f_ab.log 2>&1
cat $O/r06f_ab.log
