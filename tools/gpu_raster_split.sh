#!/bin/bash
# Sweep of SDN_RASTER_SPLIT (list length from which a tile is rasterised as four quadrants) on both template families.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
for V in ${@:-0 256 512 768 1024 1536}; do
  for M in car_like cad_like; do
    SDN_RASTER_SPLIT=$V python $R/tools/prof_geo.py --steps 30 --mesh $M --timing 2>&1 | grep PROF_GEO
  done
done | tee $O/r04_raster_split.log
