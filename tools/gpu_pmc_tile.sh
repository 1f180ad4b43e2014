#!/bin/bash
# SQ / LDS / TA counters of k_conv_tile / k_wgrad_tile (and the r03 kernels beside them) on the residual-block layer
# (tools/tile_lab.py --one): several --pmc passes, summaries -> gpurun_out/pmctile_*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  N=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmct_$N -o t -- python $R/tools/tile_lab.py --one > $O/pmctile_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmct_$N k_ > $O/pmctile_$N.json
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run sq3 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/pmctile_*.json')):
    d=json.load(open(f))
    for k,v in d.items():
        if k.startswith('_') or not isinstance(v, dict): continue
        if 'conv' in k or 'wgrad' in k: print(f.split('pmctile_')[1][:-5], k[:40], {c:round(x['mean']) for c,x in v.items()})
PY
