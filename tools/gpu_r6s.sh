#!/bin/bash
# Round 6: work conservation in the batcher (a caller blocked on a batch in flight launches what is pending if fewer than
# SYMACCEL_BATCHER_FEED groups are in flight): trait-level rate against the knob
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06s_feed.jsonl
run() { echo "# $*" >> $OUT/r06s_feed.jsonl; env "$@" | tail -1 >> $OUT/r06s_feed.jsonl; }
for rep in 1 2; do
for cfg in "0 1024" "2 1024" "3 1024" "4 1024" "6 1024" "3 4096" "4 4096" "8 2048" "3 256"; do
  set -- $cfg
  for codec in aac mp3h; do
    run SYMACCEL_BATCHER_FEED=$1 SYMACCEL_BATCHER_FEED_KB=$2 timeout 120 $B --codec $codec --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
  done
done
done
run SYMACCEL_BATCHER_FEED=3 timeout 120 $B --codec aac --streams 256 --lookahead 64 --packets 1024 --threads 16 --direct
run SYMACCEL_BATCHER_FEED=0 timeout 120 $B --codec aac --streams 256 --lookahead 64 --packets 1024 --threads 16 --direct
run SYMACCEL_BATCHER_FEED=3 timeout 120 $B --codec aac --streams 1024 --lookahead 64 --packets 1024 --threads 16 --direct
run SYMACCEL_BATCHER_FEED=0 timeout 120 $B --codec aac --streams 1024 --lookahead 64 --packets 1024 --threads 16 --direct
run SYMACCEL_BATCHER_FEED=3 timeout 120 $B --codec aac --streams 16 --lookahead 256 --packets 4096 --threads 16 --direct
run SYMACCEL_BATCHER_FEED=0 timeout 120 $B --codec aac --streams 16 --lookahead 256 --packets 4096 --threads 16 --direct
run SYMACCEL_BATCHER_FEED=3 timeout 120 $B --codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16
run SYMACCEL_BATCHER_FEED=0 timeout 120 $B --codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16
run SYMACCEL_BATCHER_FEED=3 timeout 120 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16
run SYMACCEL_BATCHER_FEED=0 timeout 120 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16
python - <<'PY'
import json
cfg=None
for l in open("gpurun_out/r06s_feed.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    print(cfg.split("timeout")[0], d["codec"], "S", d["streams"], "L", d["lookahead"], round(d["packets_per_s"]/1e6,3), "launches", d["launches"], "feed", d.get("feed_launches"), "flag_wait_ms", d["flag_wait_ms"], "fail", d["failures"])
PY
