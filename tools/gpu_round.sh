#!/bin/bash
# Runs on the GPU box: everything a round's evidence needs, for the tree as it is.
#   usage: tools/gpu_round.sh <tag> [notests]
# Writes under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_tests.log                 full `pytest -m gpu`
#   <tag>_bench.json / .err         default bench.py line
#   <tag>_{geo,tex}_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the two legs
#   <tag>_pmc_{geo,tex}_{FETCH_SIZE,WRITE_SIZE}.json   separate --pmc passes (per-kernel means, kernel names normalised)
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$2" != "notests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests exit $?" >> $O/${TAG}_tests.log
  grep -E "passed|failed|FAILED|Error" $O/${TAG}_tests.log | head -20
fi
cd /tmp; export TMPDIR=/tmp
GEO="--no-cpu-baseline --skip-textural --no-extras --steps 5 --warmup 2"
TEX="--no-cpu-baseline --skip-geometric --no-extras --textural-steps 2"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_geo -o geo -- python $R/bench.py $GEO > $O/${TAG}_prof_geo.log 2>&1
find /tmp/prof_geo -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_geo_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tex -o tex -- python $R/bench.py $TEX > $O/${TAG}_prof_tex.log 2>&1
find /tmp/prof_tex -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_tex_kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_geo_$C -o g -- python $R/bench.py $GEO > $O/${TAG}_pmc_geo_$C.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_geo_$C sdn:: $TAG > $O/${TAG}_pmc_geo_$C.json
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_tex_$C -o t -- python $R/bench.py $TEX > $O/${TAG}_pmc_tex_$C.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_tex_$C sdn:: $TAG > $O/${TAG}_pmc_tex_$C.json
done
cd $R
# the bench line last, with this run's counter files in place (bench.py reads profiles/pmc_*.json)
mkdir -p $R/profiles
cp $O/${TAG}_pmc_geo_FETCH_SIZE.json $R/profiles/pmc_FETCH_SIZE.json; cp $O/${TAG}_pmc_geo_WRITE_SIZE.json $R/profiles/pmc_WRITE_SIZE.json
cp $O/${TAG}_pmc_tex_FETCH_SIZE.json $R/profiles/pmc_tex_FETCH_SIZE.json; cp $O/${TAG}_pmc_tex_WRITE_SIZE.json $R/profiles/pmc_tex_WRITE_SIZE.json
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cut -c1-600 $O/${TAG}_bench.json
timeout 300 python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1; tail -2 $O/${TAG}_smoke.log
head -8 $O/${TAG}_geo_kernel_stats.csv | cut -c1-120
