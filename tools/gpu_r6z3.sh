#!/bin/bash
# Round 6: the busy rule of symaccel_batcher_hint (larger groups while the device has a queue) across stream counts and look-aheads
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06z3_busy.jsonl
run() { echo "# $*" >> $OUT/r06z3_busy.jsonl; env "$@" | tail -1 >> $OUT/r06z3_busy.jsonl; }
for rep in 1 2 3; do
for busy in "0 48" "4 48" "6 48" "4 64"; do
  set -- $busy
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aac --streams 16 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aac --streams 4 --lookahead 256 --packets 4096 --threads 4 --direct" "--codec aac --streams 64 --lookahead 64 --packets 1024 --threads 16 --direct" "--codec aac --streams 64 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aac --streams 16 --lookahead 64 --packets 1024 --threads 16 --direct" "--codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16"; do
    run SYMACCEL_BATCHER_BUSY_GROUPS=$1 SYMACCEL_BATCHER_BUSY_HINT_MB=$2 timeout 120 $B $args
  done
done
done
python - <<'PY'
import json
cfg=None
rows={}
for l in open("gpurun_out/r06z3_busy.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    key=(d["codec"], d["streams"], d["lookahead"])
    c=cfg.split("timeout")[0].replace("# SYMACCEL_BATCHER_BUSY_GROUPS=","g").replace("SYMACCEL_BATCHER_BUSY_HINT_MB=","mb").strip()
    rows.setdefault(key,{}).setdefault(c,[]).append(round(d["packets_per_s"]/1e6,3))
for k,v in rows.items():
    print(k, {c:x for c,x in v.items()})
PY
