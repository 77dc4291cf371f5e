#!/bin/bash
# Round 6: the copy grid cap below 64 workgroups with one chunk per group, then across stream counts / look-aheads / codecs
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
F=$OUT/r06z5_copy_grid.jsonl
: > $F
run() { echo "# $*" >> $F; env "$@" | tail -1 >> $F; }
for rep in 1 2; do
for cfg in "16 1" "32 1" "48 1" "64 1" "96 1" "64 2" "32 2" "256 3"; do
  set -- $cfg
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct"; do
    run SYMACCEL_BATCH_COPY_WGS=$1 SYMACCEL_BATCH_CHUNKS=$2 timeout 120 $B $args
  done
done
done
# across S and look-ahead: the old default against the two candidates
for cfg in "256 3" "64 1" "32 1"; do
  set -- $cfg
  for sl in "1 256" "4 256" "16 256" "64 256" "1024 256" "16 64" "64 64" "256 64" "1024 64"; do
    set -- $cfg $sl
    T=$3; [ $T -gt 16 ] && T=16
    P=4096; [ $3 -ge 1024 ] && P=2048
    run SYMACCEL_BATCH_COPY_WGS=$1 SYMACCEL_BATCH_CHUNKS=$2 timeout 200 $B --codec aac --streams $3 --lookahead $4 --packets $P --threads $T --direct
  done
  for args in "--codec mp3 --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aacd --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16 --direct" "--codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 --direct"; do
    run SYMACCEL_BATCH_COPY_WGS=$1 SYMACCEL_BATCH_CHUNKS=$2 timeout 200 $B $args
  done
done
python - <<'PY'
import json
cfg=None
rows={}
for l in open("gpurun_out/r06z5_copy_grid.jsonl"):
    l=l.strip()
    if l.startswith("#"): cfg=l; continue
    try: d=json.loads(l)
    except Exception: print(cfg, "->", l[:100]); continue
    a=cfg.split("timeout")
    c=a[0].replace("SYMACCEL_BATCH_","").replace("# ","")
    w=" ".join(a[1].split()[2:]).replace("--direct","").replace("--threads 16","")
    rows.setdefault(w,{}).setdefault(c,[]).append(round(d["packets_per_s"]/1e6,3))
for w,v in rows.items(): print(w, v)
PY
