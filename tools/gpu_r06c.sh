#!/bin/bash
# r06c: axis-aligned wave-shared passes of k_raster_tiles + the fused face set-up -- bit-exact gates, then product vs lab builds (fill
# threshold, span threshold, the r05 walk) on both synthetic families and real templates; SDN_MAPS_FUSED_SETUP=0 = the r05 launches
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_cad_golden.py tests/test_gpu_k1_coverage.py tests/test_gpu_derender3d.py tests/test_gpu_dropin.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/r06c_tests.log 2>&1; echo "tests exit $?" >> $O/r06c_tests.log
tail -3 $O/r06c_tests.log
( for M in cad_like car_like real:2 real; do
    for L in product r06_noalign r06_fill4 r06_fill6 r06_fill7 r06_span768 r06_span1025 product; do
      if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
      python tools/prof_geo.py --steps 40 --mesh $M --timing $A 2>/dev/null | grep -E "PROF_GEO"  | tr '\n' ' '; echo
    done
    SDN_MAPS_FUSED_SETUP=0 python tools/prof_geo.py --steps 40 --mesh $M --timing 2>/dev/null | grep -E "PROF_GEO"  | tr '\n' ' '; echo " (SDN_MAPS_FUSED_SETUP=0)"
  done ) > $O/r06c_raster_sweep2.log 2>&1
cat $O/r06c_raster_sweep2.log
