#!/bin/bash
# Build a tuned variant of the library here (hipcc cross-compiles) for an alternating A/B on the GPU box:
#   bash tools/build_variant.sh <name> SYMACCEL_TUNE_X=1 [SYMACCEL_TUNE_Y=2 ...]   ->  build_ab/<name>.so (+ its flags stamp)
# then: gpurun -- 'bash tools/gpu_ab_libs.sh <tag> <workload> <reps> symphonia_amd/libsymaccel.so build_ab/<name>.so ...'
set -e
NAME=$1; shift
mkdir -p build_ab
rm -f symphonia_amd/build/tuned/libsymaccel.so
env "$@" python -m symphonia_amd.build > /dev/null
cp symphonia_amd/build/tuned/libsymaccel.so build_ab/$NAME.so
cp symphonia_amd/build/tuned/libsymaccel.so.flags.json build_ab/$NAME.so.flags.json
echo "build_ab/$NAME.so: $*"
