# Lab: which source's SLP-formed packed operands break the frame gradient beside an MFMA kernel (tests/test_gpu_packed_math.py, N runs each)
cd ${GRAFT_REPO_ROOT:-.}
for L in $(ls lab/slp_*.so) product; do
  for i in 1 2 3 4 5; do
    if [ $L = product ]; then R=$(timeout 300 python -m pytest tests/test_gpu_packed_math.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1)
    else R=$(timeout 300 python tools/lab/pytest_with_lib.py $L tests/test_gpu_packed_math.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError: \{" | tr '\n' ' ' | cut -c1-260); fi
    echo "$L run $i: $R"
  done
done
