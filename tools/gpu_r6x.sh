#!/bin/bash
# Round 6: would the copy ENGINES carry the batcher's bulk planes faster than the copy kernels?  Submissions large enough that a plane
# is a worthwhile engine copy by itself (look-ahead 1024: 8 MiB per plane), SYMACCEL_BATCH_DMA_KB on / off
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06x_dma.jsonl
run() { echo "# $*" >> $OUT/r06x_dma.jsonl; env "$@" | tail -1 >> $OUT/r06x_dma.jsonl; }
for rep in 1 2; do
for dma in 0 1024; do
  run SYMACCEL_BATCH_DMA_KB=$dma timeout 120 $B --codec aac --streams 64 --lookahead 1024 --packets 16384 --threads 16 --direct
  run SYMACCEL_BATCH_DMA_KB=$dma timeout 120 $B --codec aac --streams 256 --lookahead 1024 --packets 8192 --threads 16 --direct
  run SYMACCEL_BATCH_DMA_KB=$dma timeout 120 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
  run SYMACCEL_BATCH_DMA_KB=$dma timeout 120 $B --codec aac --streams 256 --lookahead 1024 --packets 8192 --threads 16 --direct --lanes 1
  run SYMACCEL_BATCH_DMA_KB=$dma timeout 120 $B --codec mp3h --streams 256 --lookahead 1024 --packets 8192 --threads 16 --direct
done
done
python - <<'PY'
import json
cfg=None
for l in open("gpurun_out/r06x_dma.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    print(cfg.split("timeout")[0], d["codec"], "S", d["streams"], "L", d["lookahead"], "lanes", d["lanes"], round(d["packets_per_s"]/1e6,3), "launches", d["launches"], "GB/s", d["GBps_each_way"], "api_ms", d["launch_api_ms"], "fail", d["failures"])
PY
