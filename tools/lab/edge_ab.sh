# Lab: edge-kernel time (hipEvent slot: k_edge_scan_sil + k_edge_rows + k_chunk_sum) and frame step, product and lab/*.so, several meshes
cd ${GRAFT_REPO_ROOT:-.}
for M in ${@:-cad_like real real:5 real:2 car_like}; do for L in product $(ls lab/*.so 2>/dev/null | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
  python tools/prof_geo.py --steps 40 --mesh $M --timing $A 2>/dev/null | grep -E "PROF_GEO" | sed 's/(lib.*)//' | tr '\n' ' '; echo " $L"
done; done
