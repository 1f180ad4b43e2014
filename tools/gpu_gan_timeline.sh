#!/bin/bash
# rocprofv3 kernel trace of K multi-stream GAN steps -> tools/gan_timeline.py.   usage: tools/gpu_gan_timeline.sh <tag> [steps]
TAG=$1; K=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl_gan
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_gan -o gan -- python $R/tools/prof_gan.py --steps $K > $O/${TAG}_tl_gan.log 2>&1
grep PROF_GAN $O/${TAG}_tl_gan.log
F=$(find /tmp/tl_gan -name '*kernel_trace.csv' | head -1)
head -2 $F | cut -c1-400
python $R/tools/gan_timeline.py $F $K | tee $O/${TAG}_gan_timeline.txt
