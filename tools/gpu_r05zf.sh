#!/bin/bash
# r05zf: k_raster_tiles with parts switched off (lab builds; wrong pictures, timing only): where its vector instructions go
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
( for M in cad_like car_like; do
    for L in sk0 sk1 sk2 sk3 sk4 sk8 sk16 sk0; do
      python tools/prof_geo.py --steps 40 --mesh $M --timing --lib lab/$L.so 2>/dev/null | grep PROF_GEO_TIMING
    done
  done ) > $O/r05zf_raster_parts.log 2>&1
cat $O/r05zf_raster_parts.log
