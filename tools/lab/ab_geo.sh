# Lab: GPU tests of the geometric side on the product library, then the frame step (per-kernel table) on product and lab/*.so
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_derender_golden.py tests/test_gpu_derender3d.py tests/test_gpu_dropin.py tests/test_gpu_renderer.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/lab_tests.log 2>&1; tail -3 $O/lab_tests.log
for L in product $(ls lab/*.so | xargs -n1 basename | sed 's/\.so$//') product $(ls lab/*.so | xargs -n1 basename | sed 's/\.so$//'); do
  if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
  python tools/prof_geo.py --steps 40 --mesh cad_like --timing $A 2>/dev/null | grep -E "PROF_GEO " | tr '\n' ' '; echo " $L"
done
