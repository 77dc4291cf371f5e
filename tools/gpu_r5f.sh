#!/bin/bash
# Round 5, sixth GPU call: full GPU suite; the decoders line with larger chunks + Vorbis; FLAC with unpredicated tiles
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 12 > $OUT/r05f_gputest.log
cat $OUT/r05f_gputest.log
timeout 900 python bench.py --workload decoders > $OUT/r05f_decoders.json 2> $OUT/r05f_decoders.err
echo "decoders rc=$?"; tail -2 $OUT/r05f_decoders.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f_decoders.json"))["decoders"]
for r in d["sweep"]:
    g=r["gpu_batcher"]; print(r["streams"], r["threads"], round(g.get("packets_per_s",0)), round(r["cpu_port_packets_per_s"]), g.get("launches"), g.get("kernel_launches"))
for k in ("mp3_int16_S256","mp3_f32_S256","vorbis_8ch_S64"): print(k, d.get(k,{}).get("packets_per_s"), d.get(k,{}).get("GBps_each_way"))
PY
for T in 1 4 16; do $REPO/symphonia_amd/build/decoders_bench --codec aac --streams 256 --lookahead 64 --packets 256 --threads $T; done 2>&1 | cut -c1-330
timeout 600 python bench.py --workload flac --no-others --no-cpu-baseline --no-copy-ceiling > $OUT/r05f_bench_flac.json 2> $OUT/r05f_bench_flac.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05f_bench_flac.json").read().strip().splitlines()[-1])
print("flac", d["ms_per_step"], d["roofline"]["frac"], d.get("verified",{}).get("mismatches"))
PY
