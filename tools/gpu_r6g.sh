#!/bin/bash
# Round 6: is the duplex overlap lost to stream -> hardware-queue sharing?  lanes x copy-grid cap x GPU_MAX_HW_QUEUES
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=symphonia_amd/build/decoders_bench
: > $OUT/r06g_decoders.jsonl
for q in "" 8; do
for lanes in 1 2; do
for wgs in 0 64 256; do
  for codec in aac; do
    echo "{\"hwq\": \"$q\", \"lanes_cfg\": $lanes, \"wgs\": $wgs}" >> $OUT/r06g_decoders.jsonl
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    SYMACCEL_BATCH_COPY_WGS=$wgs timeout 300 $B --codec $codec --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --lanes $lanes | tee -a $OUT/r06g_decoders.jsonl
  done
done
done
done
