#!/bin/bash
# bash tools/gpu_ab_libs_w.sh <tag> "<workloads>" <reps> lib1.so lib2.so ... : tools/gpu_ab_libs.sh over several bench.py workloads
TAG=$1; WS=$2; REPS=$3; shift 3
for w in $WS; do bash tools/gpu_ab_libs.sh $TAG $w $REPS "$@"; done
