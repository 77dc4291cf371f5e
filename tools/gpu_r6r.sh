#!/bin/bash
# Round 6: chunks per group (gather(c+1) || scatter(c-1) inside one group is the only place where the two directions of the link are
# paired by construction): trait-level rate against SYMACCEL_BATCH_CHUNKS / SYMACCEL_BATCH_CHUNK_MIN_KB / --flush-mb
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06r_chunks.jsonl
run() { echo "# $*" >> $OUT/r06r_chunks.jsonl; env "$@" | tail -1 >> $OUT/r06r_chunks.jsonl; }
for rep in 1 2; do
for cfg in "3 8192 0" "6 4096 0" "8 2048 0" "12 2048 0" "16 1024 0" "8 4096 128" "16 2048 128" "6 8192 128" "3 8192 128"; do
  set -- $cfg
  for codec in aac mp3h; do
    run SYMACCEL_BATCH_CHUNKS=$1 SYMACCEL_BATCH_CHUNK_MIN_KB=$2 timeout 120 $B --codec $codec --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct $( [ $3 != 0 ] && echo --flush-mb $3 )
  done
done
done
python - <<'PY'
import json
cfg=None
for l in open("gpurun_out/r06r_chunks.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    print(cfg.split("timeout")[0], d["codec"], round(d["packets_per_s"]/1e6,3), "launches", d["launches"], "kernel_launches", d["kernel_launches"], "flag_wait_ms", d["flag_wait_ms"])
PY
