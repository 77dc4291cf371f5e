#!/bin/bash
# Round 5: ALAC with the multiply / multiply-add form of the adaptive update -- GPU parity, then the bench line (twice)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_alac.py tests/test_alac_packets.py tests/test_gpu_parity.py tests/test_rust_adapters.py tests/test_product_vs_reference_text.py -m gpu -x -q -k "alac or Alac or flac_alac" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --workload alac --no-cpu-baseline --no-copy-ceiling --repeats 3 2> $OUT/r05q_alac.err > $OUT/r05q_bench_alac_$i.json
python - $OUT/r05q_bench_alac_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("alac ms", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), "frac", round(d["roofline"]["frac"],4), "verified", d.get("verified"))
PY
done
