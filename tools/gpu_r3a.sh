#!/bin/bash
# round 3, first GPU call: the GPU suite, then the round's profiles (tools/gpu_round.sh) with the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r03a.log 2>&1
echo "gpu tests rc=$?"; tail -n 3 gpurun_out/gputest_r03a.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
bash tools/gpu_round.sh r03a
