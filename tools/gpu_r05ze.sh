#!/bin/bash
# r05ze: SMALL_AREA / SPAN_AREA of k_raster_tiles re-swept on the r05 tree (lab builds, tools/build_lab_variant.sh), both mesh families
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
( for M in cad_like car_like; do
    for L in product sa24 sa40 sa48 sp384 sp768 sp1024 product; do
      if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
      python tools/prof_geo.py --steps 40 --mesh $M --timing $A 2>/dev/null | grep PROF_GEO_TIMING
    done
  done ) > $O/r05ze_raster_sweep.log 2>&1
cat $O/r05ze_raster_sweep.log
