#!/bin/bash
# Per-step kernel table of the GAN train step: rocprofv3 kernel stats of tools/prof_gan.py (K identical steps, single stream),
# every total divided by K.   usage: tools/gpu_prof_gan.sh <tag> [steps]
TAG=$1; K=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_gan
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gan -o gan -- python $R/tools/prof_gan.py --steps $K --single-stream > $O/${TAG}_prof_gan.log 2>&1
find /tmp/prof_gan -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_gan_kernel_stats.csv \;
grep PROF_GAN $O/${TAG}_prof_gan.log
python $R/tools/gan_step_table.py $O/${TAG}_gan_kernel_stats.csv $K | tee $O/${TAG}_gan_step_table.md
