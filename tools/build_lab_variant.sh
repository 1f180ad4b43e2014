#!/bin/bash
# lab/<name>.so = the tree's library with one source recompiled under extra -D flags (A/B runs on one GPU box).
#   usage: tools/build_lab_variant.sh <name> <source.hip> <flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd)
N=$1; SRC=$2; shift 2
mkdir -p $R/lab/obj_$N
C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=${FPC:-fast}"
# (the product builds everything but conv_*.hip with -fno-slp-vectorize: see csrc/Makefile)
case $SRC in conv_*) ;; *) if [ "${SLP:-0}" != 1 ]; then C="$C -fno-slp-vectorize"; fi ;; esac   # SLP=1: lab build with the vectoriser on
/opt/rocm/bin/hipcc $C "$@" -c $R/3d-sdn_amd/csrc/$SRC -o $R/lab/obj_$N/${SRC%.hip}.o
OBJS=$(ls $R/3d-sdn_amd/lib/obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/lab/obj_$N/${SRC%.hip}.o -o $R/lab/$N.so
echo built $R/lab/$N.so
