cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_packed_math.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
for i in 1 2; do python tools/prof_geo.py --steps 40 --mesh cad_like --timing 2>/dev/null | grep PROF_GEO | tr '\n' ' '; echo; done
python tools/prof_geo.py --steps 40 --mesh real --timing 2>/dev/null | grep PROF_GEO | tr '\n' ' '; echo
python tools/prof_geo.py --steps 40 --mesh car_like --timing 2>/dev/null | grep PROF_GEO | tr '\n' ' '; echo
