# Lab: the GAN step (and the serial per-layer totals) with the product library and lab/*.so
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out; mkdir -p $O
for L in product $(ls lab/*.so | xargs -n1 basename | sed 's/\.so$//') product $(ls lab/*.so | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $L = product ]; then unset LAB_LIB; else export LAB_LIB=$PWD/lab/$L.so; fi
  python - <<PY
import os, sys, time
sys.path[:0] = ['.', '3d-sdn_amd']
import sdn_hip
if os.environ.get('LAB_LIB'): sdn_hip.LIB_PATH = os.environ['LAB_LIB']
import torch, bench
t = bench.textural_leg(torch.device('cuda', 0), 8, 1, 1)
print('%-14s GAN step %.2f ms   single-stream %.2f ms  issued %.3f' % ('$L', t['ms_per_step'], t['roofline'].get('single_stream', {}).get('ms_per_step', 0), t['roofline'].get('single_stream', {}).get('issued_frac', 0)), flush=True)
PY
done 2>&1 | grep "GAN step"
