# Lab: the product library beside lab builds (lab/*.so) on the frame step -- parity tests of the product, then alternating timings.
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_cad_golden.py tests/test_gpu_derender3d.py tests/test_gpu_dropin.py tests/test_gpu_k1_coverage.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/lab_tests.log 2>&1; tail -3 $O/lab_tests.log
for M in cad_like real car_like; do for L in product $(ls lab/*.so | xargs -n1 basename | sed 's/\.so$//') product $(ls lab/*.so | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
  python tools/prof_geo.py --steps 40 --mesh $M --timing $A 2>/dev/null | grep -E "PROF_GEO" | tr '\n' ' '; echo
done; done > $O/lab_ab.log 2>&1; cat $O/lab_ab.log
