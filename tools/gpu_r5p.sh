#!/bin/bash
# Round 5 (scratch): what the dct32 pass costs in the MP3 walk -- ablations, timing only (the ablated builds fail the bench's verification: the log line is read)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2; do
for v in ${VARIANTS:-product dct_1 dct_2}; do
  if [ $v = product ]; then unset SYMACCEL_LIB; else export SYMACCEL_LIB=$PWD/symphonia_amd/build/tuned/$v/libsymaccel.so; fi
  for w in ${WORKLOADS:-mp3 mp3q}; do
  timeout 200 python bench.py --workload $w --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 3 2> $OUT/r05p_$v.err > /dev/null
  echo "$v $w $(grep -o 'repeats done.*' $OUT/r05p_$v.err)"
  done
done
done
