#!/bin/bash
# Development tool (GPU box): power, clocks and temperature while one workload runs back to back for a few seconds.
#   bash tools/power_probe.sh <workload> [library]
W=$1; LIB=${2:-symphonia_amd/libsymaccel.so}
export TMPDIR=/tmp
echo "== idle"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (junction|edge|memory)" | head -12
SYMACCEL_LIB=$LIB python bench.py --workload $W --steps 20000 --warmup 50 --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3; do echo "== under $W, sample $i"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (junction|edge|memory)" | head -12; sleep 1; done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
wait $BP
python -c "import json; d=json.load(open('/tmp/pp_bench.json')); print('$W', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"
