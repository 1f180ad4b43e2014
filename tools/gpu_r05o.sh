#!/bin/bash
# r05o: the row kernel of sdn_conv_wgrad_narrow with explicit v_pk_fma_f32 broadcasts (parity + timing)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r05o}
timeout 600 python -m pytest tests/test_gpu_wgrad_narrow.py -q --tb=short -p no:cacheprovider > $O/${T}_tests.log 2>&1; tail -15 $O/${T}_tests.log
( timeout 300 python tools/narrow_lab.py; SDN_WGRAD_NARROW_ROW=0 timeout 300 python tools/narrow_lab.py ) > $O/${T}_narrow_lab.log 2>&1; cat $O/${T}_narrow_lab.log
