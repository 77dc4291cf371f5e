#!/bin/bash
# Round 6: lanes again, now that the copy launches are capped at 64 workgroups (one lane was as good as two with 256)
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
F=$OUT/r06z15_lanes.jsonl
: > $F
run() { echo "# $*" >> $F; env "$@" | tail -1 >> $F; }
for rep in 1 2; do
for cfg in "1 2" "2 2" "3 2" "4 2" "6 2" "3 1" "4 1" "4 3"; do
  set -- $cfg
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --via-registry" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --via-registry" "--codec aac --streams 64 --lookahead 64 --packets 4096 --threads 16 --direct --via-registry"; do
    run SYMACCEL_BATCHER_LANES=$1 SYMACCEL_BATCH_CHUNKS=$2 timeout 120 $B $args
  done
done
done
python - <<'PY'
import json
cfg=None
rows={}
for l in open("gpurun_out/r06z15_lanes.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    a=cfg.split("timeout")
    c=a[0].replace("SYMACCEL_BATCHER_","").replace("SYMACCEL_BATCH_","").replace("# ","")
    w=" ".join(a[1].split()[2:8])
    rows.setdefault(w,{}).setdefault(c,[]).append((round(d["packets_per_s"]/1e6,3), d["lanes"]))
for w,v in rows.items(): print(w, v)
PY
