#!/bin/bash
# Round 6: the GPU suite again after the chunk-count assertion was corrected, smoke(), and the default line with the final bench.py
OUT=$PWD/gpurun_out; mkdir -p $OUT/profiles_r06zz; export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 5 > $OUT/profiles_r06zz/r06zz2_gputest.log
cat $OUT/profiles_r06zz/r06zz2_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/profiles_r06zz/r06zz2_default_bench.json 2> $OUT/r06zz2_default.err; echo "default rc=$?"; tail -2 $OUT/r06zz2_default.err
cut -c1-300 $OUT/profiles_r06zz/r06zz2_default_bench.json
