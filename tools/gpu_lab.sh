#!/bin/bash
# One-lease lab runs of round 6, folded into one script (the r05 one-offs are in the git history; their outputs are under
# profiles/ and described in profiles/HISTORY.md).   usage: tools/gpu_lab.sh <what> [args]
#   baseline        GPU tests, geometric bench line with the real-mesh block, tile statistics of the six real templates
#   tex             batch-4 oracle gates, per-layer times incl. the encoder (serial), textural bench line
#   raster-sweep    raster / renderer parity tests, then product vs lab builds of raster_fwd.hip (lab/*.so, tools/build_lab_variant.sh)
#   update-ab       train-step gates, then the GAN step with SDN_UPDATE_STREAM=1 / 0
#   head-race       tools/lab/head_race*.py: the head weight gradient beside an MFMA kernel on another stream (packed-FMA erratum)
#   whead           tests of sdn_conv_wgrad_head_mfma, then its time beside sdn_conv_wgrad_narrow at the product's shapes
#   whead-ab        train-step / full-size / encoder gates, then the GAN step with SDN_WGRAD_HEAD=0 / 1 / 2
#   timeline [K]    rocprofv3 kernel trace of K GAN steps -> tools/gan_timeline.py (idle time, overlap, per-millisecond Gantt chart)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
W=$1; shift
case $W in
baseline)
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -x > $O/lab_tests.log 2>&1; tail -3 $O/lab_tests.log
  timeout 600 python bench.py --skip-textural --no-cpu-baseline > $O/lab_bench_geo.json 2> $O/lab_bench_geo.err; cut -c1-300 $O/lab_bench_geo.json
  timeout 600 python tools/tile_stats.py cad_like real:0 real:1 real:2 real:3 real:4 real:5 > $O/lab_tile_stats.log 2>&1; tail -5 $O/lab_tile_stats.log ;;
tex)
  timeout 900 python -m pytest tests/test_gpu_textural_fullsize.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "batch4" -s > $O/lab_tests.log 2>&1
  grep -E "passed|failed|batch-4|batch 4|D\(3" $O/lab_tests.log | head
  SDN_WGRAD_STREAM=0 SDN_D_STREAMS=0 timeout 400 python tests/gpu_layer_times.py > $O/lab_layer_times_serial.log 2>&1; grep -E "^totals|^====" $O/lab_layer_times_serial.log
  timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras > $O/lab_bench_tex.json 2> $O/lab_bench_tex.err; cut -c1-400 $O/lab_bench_tex.json ;;
raster-sweep)
  timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_renderer.py tests/test_gpu_cad_golden.py tests/test_gpu_k1_coverage.py tests/test_gpu_derender3d.py tests/test_gpu_dropin.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/lab_tests.log 2>&1; tail -3 $O/lab_tests.log
  ( for M in cad_like car_like real:2 real; do
      for L in product $(ls lab/*.so 2>/dev/null | xargs -n1 basename | sed 's/\.so$//') product; do
        if [ $L = product ]; then A=""; else A="--lib lab/$L.so"; fi
        python tools/prof_geo.py --steps 40 --mesh $M --timing $A 2>/dev/null | grep -E "PROF_GEO" | tr '\n' ' '; echo
      done
      SDN_MAPS_FUSED_SETUP=0 python tools/prof_geo.py --steps 40 --mesh $M --timing 2>/dev/null | grep -E "PROF_GEO" | tr '\n' ' '; echo " (SDN_MAPS_FUSED_SETUP=0)"
    done ) > $O/lab_raster_sweep.log 2>&1; cat $O/lab_raster_sweep.log ;;
update-ab)
  timeout 900 python -m pytest tests/test_gpu_trainstep.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/lab_tests.log 2>&1; tail -3 $O/lab_tests.log
  for U in 1 0 1 0; do
    SDN_UPDATE_STREAM=$U timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras --textural-steps 8 > $O/lab_bench_tex_u$U.json 2> /dev/null
    echo "SDN_UPDATE_STREAM=$U $(cut -c1-60 $O/lab_bench_tex_u$U.json)"
  done ;;
head-race)
  python tools/lab/head_race3.py 2>&1 | grep -E "library|main stream"
  python tools/lab/head_race.py 192 624 2>&1 | grep -E "HEAD_WIDE|cout|by output" ;;
whead)
  timeout 900 python -m pytest tests/test_gpu_wgrad_head.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/lab_tests.log 2>&1; tail -12 $O/lab_tests.log
  timeout 300 python tools/lab/whead_time.py > $O/lab_whead_time.log 2>&1; cat $O/lab_whead_time.log ;;
whead-ab)
  timeout 1200 python -m pytest tests/test_gpu_trainstep.py tests/test_gpu_textural_fullsize.py tests/test_gpu_encoder.py -m gpu -q --tb=short -rf -p no:cacheprovider > $O/lab_tests.log 2>&1; tail -5 $O/lab_tests.log
  for U in 0 1 2 0 1 2; do
    SDN_WGRAD_HEAD=$U timeout 600 python bench.py --skip-geometric --no-cpu-baseline --no-extras --textural-steps 8 > $O/lab_bench_tex_wh$U.json 2> /dev/null
    echo "SDN_WGRAD_HEAD=$U $(cut -c1-60 $O/lab_bench_tex_wh$U.json)"
  done ;;
timeline)
  bash tools/gpu_gan_timeline.sh lab ${1:-3} ;;
*) echo "usage: tools/gpu_lab.sh baseline|tex|raster-sweep|update-ab|head-race|whead|whead-ab|timeline" ;;
esac
