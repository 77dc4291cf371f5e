#!/bin/bash
# Round 5: buffer placement against the AAC headline's launch time (one process, offsets in alternation), three processes
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 200 python tools/aac_offsets.py > $OUT/r05n_aac_offsets_$i.json 2> $OUT/r05n_aac_offsets_$i.err; echo "rc=$?"
  cat $OUT/r05n_aac_offsets_$i.json; tail -3 $OUT/r05n_aac_offsets_$i.err
done
