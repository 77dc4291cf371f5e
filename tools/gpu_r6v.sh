#!/bin/bash
# Round 6: the launches of the trait-level pipeline one after the other (one lane): who runs beside whom
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
B=$REPO/symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$REPO/symphonia_amd:$LD_LIBRARY_PATH
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt_r06v -o p -- $B --codec aac --streams 256 --lookahead 256 --packets 2048 --threads 16 --direct --lanes 1 2>/dev/null | tail -1 | cut -c1-200
python $REPO/tools/copy_timeline.py $(find $OUT/kt_r06v -name '*.db' | head -1) dump > $OUT/r06v_timeline_dump.txt
rm -rf $OUT/kt_r06v
head -100 $OUT/r06v_timeline_dump.txt
