python -m pytest tests/test_row_stride.py tests/test_batcher_kinds.py tests/test_flac_packets.py tests/test_alac_packets.py tests/test_alac.py -m gpu -x -q 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 4 > gpurun_out/r06zz32_tests.log; cat gpurun_out/r06zz32_tests.log
for w in flac flacp alac alacp; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-others --no-cpu-baseline > gpurun_out/r06zz32_bench_$w.json 2> gpurun_out/r06zz32_bench_$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06zz32_bench_$w.json").read().strip().splitlines()[-1])
print("$w", d["ms_per_step"], d["roofline"]["frac"], d.get("verified"))
PY
done
E=symphonia_amd/build/decoders_bench
for rep in 1 2 3; do
for pad in 1 0; do
  SYMACCEL_BATCH_ROW_PAD=$pad $E --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 --via-registry | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flac S256 la64 pad=$pad', d['packets_per_s'], d['failures'] if 'failures' in d else '')" 
done; done >> gpurun_out/r06zz32_trait_flac_ab.txt 2>&1
cat gpurun_out/r06zz32_trait_flac_ab.txt
