# Lab: the frame step under an environment switch, alternating.   usage: tools/lab/env_ab.sh VAR "v1 v2 ..." [mesh ...]
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out; mkdir -p $O
VAR=$1; VALS=$2; shift 2; MESHES=${@:-cad_like real car_like}
for M in $MESHES; do for R in 1 2; do for V in $VALS; do
  echo -n "$VAR=$V  "; env $VAR=$V python tools/prof_geo.py --steps 40 --mesh $M --timing 2>/dev/null | grep -E "PROF_GEO" | tr '\n' ' '; echo
done; done; done 2>&1 | tee $O/lab_env_ab.log
