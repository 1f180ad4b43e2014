#!/bin/bash
# SQ / LDS / L2 counters of k_conv_wgrad on the dominant shapes (tools/wgrad_lab.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  N=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcw_$N -o t -- python $R/tools/wgrad_lab.py > $O/pmcwgrad_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmcw_$N k_conv_wgrad > $O/pmcwgrad_$N.json
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
run grbm GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/pmcwgrad_*.json')):
    d=json.load(open(f))
    for k,v in d.items():
        print(f.split('_')[-1], k[:50], {c:(round(x['mean']/1e6,2),x['dispatches']) for c,x in v.items()})
PY
