#!/bin/bash
# Round 6: where a submission's time goes behind the trait: commit -> launch, launch -> completion seen, waits that found the word missing
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
: > $OUT/r06u_latency.jsonl
for args in "--streams 256 --lookahead 256 --packets 4096 --threads 16" "--streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes 1" "--streams 16 --lookahead 256 --packets 4096 --threads 16" "--streams 256 --lookahead 256 --packets 4096 --threads 32" "--streams 256 --lookahead 256 --packets 4096 --threads 64" "--streams 256 --lookahead 256 --packets 4096 --threads 8" "--streams 1024 --lookahead 256 --packets 4096 --threads 16"; do
  timeout 120 $B --codec aac --direct $args | tail -1 >> $OUT/r06u_latency.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06u_latency.jsonl"):
    d=json.loads(l)
    print("S", d["streams"], "T", d["threads"], "lanes", d["lanes"], round(d["packets_per_s"]/1e6,3), "M/s  launches", d["launches"], "subs/launch", round(d["decoder_batches"]/max(1,d["launches"]),2), "commit->launch ms", d["commit_to_launch_ms_per_submission"], "launch->done ms", d["launch_to_done_ms"], "waits", d["waits"], "blocked", d["waits_blocked"], "flag_wait_ms/thread", round(d["flag_wait_ms"]/d["threads"],1), "of", round(d["seconds"]*1e3,1))
PY
