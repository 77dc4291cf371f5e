#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python bench.py > $OUT/r05j_default_bench.json 2> $OUT/r05j_default_bench.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05j_default_bench.json"))
print(d["value"], d["roofline"]["frac"])
h=d["host_to_host"]; print({k:h[k] for k in ("ms","GBps_each_way")}, h["from_coded_spectra"])
print(d["host_to_host_mp3"]["int16_samples"]["ms"], d["host_to_host_mp3"]["f32_spectra"]["ms"])
PY
tail -3 $OUT/r05j_default_bench.err
