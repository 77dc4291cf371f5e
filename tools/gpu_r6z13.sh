#!/bin/bash
# Round 6: the ILP scheduling strategy on the other issue-bound kernels (mp3.hip, alac.hip, flac.hip, vorbis.hip, vorbis_wave.hip): product against build_ab/ilp_all
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
show() { python - $1 $2 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4) if d.get("repeats") else None, "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
}
for w in mp3 mp3q flac alac vorbis vorbisf; do for v in product ilp_all product ilp_all; do
  L=$PWD/build_ab/$v/libsymaccel.so; [ $v = product ] && L=$PWD/symphonia_amd/libsymaccel.so
  SYMACCEL_LIB=$L timeout 300 python bench.py --workload $w --no-others --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 --steps 64 2> $OUT/r06z13.err > $OUT/r06z13_bench_${w}_$v.json; show $OUT/r06z13_bench_${w}_$v.json ${w}_$v
done; done
