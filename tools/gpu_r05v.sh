#!/bin/bash
# r05v: SQ counters of the frame step's kernels on the r05 tree, safe rule and K1 (two --pmc passes each; kernel-trace + pmc only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # name, extra prof_geo flags, counters...
  N=$1; F=$2; shift; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcg_$N -o t -- python $R/tools/prof_geo.py --steps 6 --mesh cad_like $F > $O/r05v_pmcgeo_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmcg_$N sdn:: r05v > $O/r05v_pmcgeo_$N.json
}
run sq1 "" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 "" SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE
run k1_sq1 "--k1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run k1_sq2 "--k1" SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/r05v_pmcgeo_*.json')):
    d=json.load(open(f))
    for k,v in d.items():
        if k.startswith('_') or not isinstance(v, dict): continue
        if any(s in k for s in ('raster_tiles','edge_scan_sil','edge_rows')): print(f.split('pmcgeo_')[1][:-5], k[:34], {c:round(x['mean']) for c,x in v.items()})
PY
