#!/bin/bash
# r05n: the row kernel of sdn_conv_wgrad_narrow (parity + timing against the column kernel), and the configs[4] pipeline
# with / without the MFMA head kernel (r05m measured 8.9 ms per frame against 6.4-7.5 before)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_wgrad_narrow.py -q --tb=short -p no:cacheprovider > $O/r05n_tests.log 2>&1; tail -15 $O/r05n_tests.log
( timeout 300 python tools/narrow_lab.py; SDN_WGRAD_NARROW_ROW=0 timeout 300 python tools/narrow_lab.py ) > $O/r05n_narrow_lab.log 2>&1; cat $O/r05n_narrow_lab.log
( timeout 300 python tools/pipe_lab.py; SDN_TILE_KERNELS=wfhdp timeout 300 python tools/pipe_lab.py ) > $O/r05n_pipe_lab.log 2>&1; grep "run" $O/r05n_pipe_lab.log
