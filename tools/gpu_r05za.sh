#!/bin/bash
# the drop-in optimisation loop (configs[2]) after trimming render()'s small torch launches: parity tests, launch attribution, loop time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r05za}
timeout 900 python -m pytest tests/test_gpu_derender3d.py tests/test_gpu_dropin.py tests/test_gpu_pipeline_e2e.py -q --tb=short -p no:cacheprovider 2>&1 | tail -5
python tools/attribute_torch_ops.py $O/${T}_torch_ops_opt.txt opt 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | grep "launches-worth"
python - <<PY 2>&1 | grep -v "Warning\|warn\|amdgpu.ids"
import sys, torch
sys.path.insert(0, '$R')
import bench
for k in range(3):
    r = bench.derender3d_loop(torch.device('cuda', 0))
    print('optimisation %.2f ms (%.3f per iteration, %.0f objects/s)  inference %.2f  train step %.2f' % (r['optimisation_ms'], r['optimisation_ms_per_iteration'], r['optimisation_objects_per_s'], r['inference_ms'], r['train_step_ms']))
PY
