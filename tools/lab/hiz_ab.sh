# Lab: k_raster_tiles with the hierarchical-z cull off / always / per tile (thresholds: product + lab/hizmin*.so)
cd ${GRAFT_REPO_ROOT:-.}
for M in cad_like real real:2 car_like; do
  for R in 1 2; do
    for C in "0 product" "1 product" "2 product" $(ls lab/hizmin*.so 2>/dev/null | xargs -n1 basename | sed 's/\.so$//' | sed 's/^/2 /' | tr '\n' ';' | sed 's/;/" "/g'); do :; done
    for H in 0 1 2; do echo -n "HIZ=$H product      "; SDN_RASTER_HIZ=$H python tools/prof_geo.py --steps 40 --mesh $M --timing 2>/dev/null | grep PROF_GEO_TIMING; done
    for L in $(ls lab/hizmin*.so 2>/dev/null); do echo -n "HIZ=2 $(basename $L) "; SDN_RASTER_HIZ=2 python tools/prof_geo.py --steps 40 --mesh $M --timing --lib $L 2>/dev/null | grep PROF_GEO_TIMING; done
  done
done
