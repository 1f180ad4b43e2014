# Lab: per-kernel times of the frame step (rocprofv3 kernel stats) for the product library and every lab/*.so
#   usage: tools/lab/kernel_ab.sh "<kernel name pattern (egrep)>" [mesh]
cd ${GRAFT_REPO_ROOT:-.}; PAT=$1; MESH=${2:-cad_like}; export TMPDIR=/tmp
for L in product $(ls lab/*.so 2>/dev/null | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $L = product ]; then A=""; else A="--lib $PWD/lab/$L.so"; fi
  rm -rf /tmp/kab; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kab -o k -- python $OLDPWD/tools/prof_geo.py --steps 20 --mesh $MESH $A > /tmp/kab.log 2>&1 )
  echo "== $L: $(grep 'PROF_GEO ' /tmp/kab.log)"
  python - <<PY
import csv, glob, re
f = glob.glob('/tmp/kab/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(r'''$PAT''', r['Name']):
        print('   %-60s %8.1f us / launch' % (r['Name'][:60], float(r['AverageNs']) / 1e3))
PY
done
