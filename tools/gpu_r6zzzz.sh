#!/bin/bash
# Round 6, the evidence run on the final tree (ABI 8: padded FLAC / ALAC rows, flacp / alacp lines): tools/gpu_round.sh (kernel stats, HBM traffic, the GPU suite, every bench line, the default
# line) + SQ-counter passes of the kernels this round's last changes touched + the decoders line + the trait-level sweep through the registry.
bash tools/gpu_round.sh r06zzzz
bash tools/gpu_pmc.sh r06zzzz flac flacp
cp gpurun_out/r06zzzz_*_sq_counters.txt gpurun_out/profiles_r06zzzz/ 2>/dev/null
timeout 900 python bench.py --workload decoders > gpurun_out/profiles_r06zzzz/r06zzzz_decoders.json 2> gpurun_out/r06zzzz_decoders.err
echo "decoders rc=$?"
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
F=gpurun_out/profiles_r06zzzz/r06zzzz_trait_sweep.jsonl
: > $F
for rep in 1 2 3; do
  for sl in "1 256" "4 256" "16 256" "64 256" "256 256" "1024 256" "1 64" "16 64" "64 64" "256 64" "1024 64"; do
    set -- $sl; T=$1; [ $T -gt 16 ] && T=16; P=4096; [ $1 -ge 1024 ] && P=2048
    timeout 200 $B --codec aac --streams $1 --lookahead $2 --packets $P --threads $T --direct --via-registry | tail -1 >> $F
  done
  for args in "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec mp3 --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec aacd --streams 256 --lookahead 256 --packets 4096 --threads 16" "--codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16" "--codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16"; do
    timeout 200 $B $args --via-registry | tail -1 >> $F
  done
done
python - <<'PY'
import json,statistics
rows={}
for l in open("gpurun_out/profiles_r06zzzz/r06zzzz_trait_sweep.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    rows.setdefault((d["codec"],d["streams"],d["lookahead"]),[]).append(d["packets_per_s"]/1e6)
for k,v in rows.items(): print(k, "median %.3f"%statistics.median(v), ["%.3f"%x for x in v])
PY
ls gpurun_out/profiles_r06zzzz | wc -l
