#!/bin/bash
# Round 6: larger groups (longer copy kernels): the hint threshold against the trait-level rate
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06z1_hint.jsonl
run() { echo "# $*" >> $OUT/r06z1_hint.jsonl; env "$@" | tail -1 >> $OUT/r06z1_hint.jsonl; }
for rep in 1 2; do
for hint in 4 16 32 64 200; do
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --lanes 1"; do
    run SYMACCEL_BATCHER_HINT_MB=$hint timeout 120 $B $args
  done
done
done
python - <<'PY'
import json
cfg=None
for l in open("gpurun_out/r06z1_hint.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    print(cfg.split("timeout")[0], d["codec"], "lanes", d["lanes"], "->", round(d["packets_per_s"]/1e6,3), "launches", d["launches"], "subs/launch", round(d["decoder_batches"]/max(1,d["launches"]),1), "launch->done ms", d["launch_to_done_ms"])
PY
