#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 900 python bench.py --workload decoders > $OUT/r05i_decoders.json 2> $OUT/r05i_decoders.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05i_decoders.json"))["decoders"]
for key in ("sweep","sweep_lookahead_256"):
    for r in d[key]:
        g=r["gpu_batcher"]; print(key, r["streams"], r["threads"], round(g.get("packets_per_s",0)), [round(x) for x in g.get("runs_packets_per_s",[])], round(r["cpu_port_packets_per_s"]), round(r["gpu_over_cpu"],2) if r["gpu_over_cpu"] else None)
PY
