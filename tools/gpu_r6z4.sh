#!/bin/bash
# Round 6: with the busy rule in (groups of 24-40 submissions), the copy grid cap, the chunk count and the flush size again
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06z4_big_groups.jsonl
run() { echo "# $*" >> $OUT/r06z4_big_groups.jsonl; env "$@" | tail -1 >> $OUT/r06z4_big_groups.jsonl; }
for rep in 1 2; do
for cfg in "256 3 8192 0 48" "64 3 8192 0 48" "128 3 8192 0 48" "256 1 8192 0 48" "256 2 8192 0 48" "256 6 8192 0 48" "64 1 8192 0 48" "256 3 8192 128 96" "64 3 8192 128 96" "256 3 8192 256 200" "256 3 8192 32 24"; do
  set -- $cfg
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct"; do
    run SYMACCEL_BATCH_COPY_WGS=$1 SYMACCEL_BATCH_CHUNKS=$2 SYMACCEL_BATCH_CHUNK_MIN_KB=$3 SYMACCEL_BATCHER_BUSY_HINT_MB=$5 timeout 120 $B $args $( [ $4 != 0 ] && echo --flush-mb $4 )
  done
done
done
python - <<'PY'
import json
cfg=None
rows={}
for l in open("gpurun_out/r06z4_big_groups.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    c=cfg.split("timeout")[0].replace("SYMACCEL_BATCH_","").replace("SYMACCEL_BATCHER_","").replace("# ","")+(" flush "+cfg.split("--flush-mb")[1] if "--flush-mb" in cfg else "")
    rows.setdefault(c,{}).setdefault(d["codec"],[]).append(round(d["packets_per_s"]/1e6,3))
for c,v in rows.items(): print(c, v)
PY
