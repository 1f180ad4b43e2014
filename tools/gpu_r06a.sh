#!/bin/bash
# r06a: baseline of the round on a fresh box -- GPU tests, the geometric bench line with the real-mesh block, tile statistics of the
# real templates (why is mesh 2 = 3776e4d1 slow?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -x > $O/r06a_tests.log 2>&1; echo "tests exit $?" >> $O/r06a_tests.log
tail -3 $O/r06a_tests.log
timeout 600 python bench.py --skip-textural --no-cpu-baseline > $O/r06a_bench_geo.json 2> $O/r06a_bench_geo.err; cut -c1-300 $O/r06a_bench_geo.json
timeout 600 python tools/tile_stats.py cad_like real:0 real:1 real:2 real:3 real:4 real:5 > $O/r06a_tile_stats.log 2>&1; cp $O/tile_stats.json $O/r06a_tile_stats.json
tail -5 $O/r06a_tile_stats.log
