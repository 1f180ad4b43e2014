#!/bin/bash
# SQ / LDS / L2 counters of the conv kernels during one generator forward+backward (tests/gpu_layer_times.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  N=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$N -o t -- python $R/tests/gpu_layer_times.py > $O/pmcconv_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_$N k_conv_gemm > $O/pmcconv_$N.json
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/pmcconv_*.json')):
    d=json.load(open(f))
    for k,v in d.items():
        if '2, 2, 2, 2, 2' in k: print(k[:50], {c:(round(x['mean']),x['dispatches']) for c,x in v.items()})
PY
