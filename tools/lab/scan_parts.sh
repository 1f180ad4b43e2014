cd ${GRAFT_REPO_ROOT:-.}
for L in product scan_nofile scan_noin scan_neither; do
  if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
  python tools/prof_geo.py --steps 40 --mesh cad_like --timing $A 2>/dev/null | grep -E "PROF_GEO" | tr '\n' ' '; echo
done
