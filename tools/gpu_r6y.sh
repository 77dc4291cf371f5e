#!/bin/bash
# Round 6: what ONE engine copy per group and direction would give the real pipeline (SYMACCEL_BATCH_FAKE_MIRROR: the bulk planes of a chunk
# as one copy from / to a page-locked dummy -- results wrong on purpose, the harness only counts packets)
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06y_fake_mirror.jsonl
run() { echo "# $*" >> $OUT/r06y_fake_mirror.jsonl; env "$@" | tail -1 >> $OUT/r06y_fake_mirror.jsonl; }
for rep in 1 2; do
for cfg in "0 0" "1024 0" "1024 1"; do
  set -- $cfg
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --lanes 1" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aac --streams 256 --lookahead 64 --packets 1024 --threads 16 --direct" "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --flush-mb 16"; do
    run SYMACCEL_BATCH_DMA_KB=$1 SYMACCEL_BATCH_FAKE_MIRROR=$2 timeout 120 $B $args
  done
done
done
python - <<'PY'
import json
cfg=None
for l in open("gpurun_out/r06y_fake_mirror.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    print(cfg.split("timeout")[0], cfg.split("decoders_bench")[1][:90], "->", round(d["packets_per_s"]/1e6,3), "launches", d["launches"], "GB/s", d["GBps_each_way"], "api_ms", d["launch_api_ms"])
PY
