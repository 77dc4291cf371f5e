#!/bin/bash
# Round 6: two TNS filters per lane (packed arithmetic); MP3 int16 -> PCM at three / four wavefronts per SIMD; the copy grid per direction
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_aac_tools.py tests/test_aac_js_fused.py tests/test_gpu_fuzz.py tests/test_aac_packets.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
show() { python - $1 $2 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4) if d.get("repeats") else None, "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
}
timeout 600 python bench.py --workload aactns --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 2 > $OUT/r06z6_bench_aactns.json 2> $OUT/r06z6_aactns.err; show $OUT/r06z6_bench_aactns.json aactns
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06z6_aactns -o aactns -- python $OLDPWD/bench.py --workload aactns --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/prof_r06z6_aactns/aactns_results.db > $OUT/r06z6_aactns_rocprofv3.txt 2>&1; head -12 $OUT/r06z6_aactns_rocprofv3.txt | cut -c1-200; rm -rf $OUT/prof_r06z6_aactns
for v in product mp3f_12_3 mp3f_16_4 mp3f_8_2 mp3f_6_3; do
  L=$PWD/build_ab/$v/libsymaccel.so; [ $v = product ] && L=$PWD/symphonia_amd/libsymaccel.so
  SYMACCEL_LIB=$L timeout 300 python bench.py --workload mp3q --no-others --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 --steps 64 2> $OUT/r06z6_mp3q_$v.err > $OUT/r06z6_bench_mp3q_$v.json; show $OUT/r06z6_bench_mp3q_$v.json mp3q_$v
done
B=symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$PWD/symphonia_amd:$LD_LIBRARY_PATH
F=$OUT/r06z6_copy_gs.jsonl
: > $F
run() { echo "# $*" >> $F; env "$@" | tail -1 >> $F; }
for rep in 1 2; do
for cfg in "64 64" "96 48" "48 96" "64 32" "32 64" "128 64" "64 128" "96 64" "64 96" "80 80"; do
  set -- $cfg
  for args in "--codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct" "--codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct"; do
    run SYMACCEL_BATCH_COPY_WGS_G=$1 SYMACCEL_BATCH_COPY_WGS_S=$2 timeout 120 $B $args
  done
done
done
for cfg in "256 3" "64 2"; do
  set -- $cfg
  for args in "--codec aacd --streams 256 --lookahead 256 --packets 4096 --threads 16" "--codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16" "--codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16"; do
    run SYMACCEL_BATCH_COPY_WGS=$1 SYMACCEL_BATCH_CHUNKS=$2 timeout 200 $B $args
  done
done
python - <<'PY'
import json
cfg=None
rows={}
for l in open("gpurun_out/r06z6_copy_gs.jsonl"):
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
