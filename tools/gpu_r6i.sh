#!/bin/bash
# Round 6: NUMA placement of the callers (and of the page-locked slabs they allocate) against the GPU's node
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=$PWD/symphonia_amd/build/decoders_bench
{
lscpu | grep -i -E "numa|socket|model name" 
for d in /sys/class/drm/card*/device; do echo $d $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null); done
which numactl taskset
cat /sys/devices/system/node/node*/cpulist 2>/dev/null
} > $OUT/r06i_numa.txt 2>&1
cat $OUT/r06i_numa.txt
: > $OUT/r06i_decoders.jsonl
nodes=$(ls -d /sys/devices/system/node/node* | wc -l)
for n in $(seq 0 $((nodes-1))); do
  cpus=$(cat /sys/devices/system/node/node$n/cpulist)
  echo "{\"node\": $n, \"cpus\": \"$cpus\"}" >> $OUT/r06i_decoders.jsonl
  taskset -c $cpus $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct | tee -a $OUT/r06i_decoders.jsonl
  taskset -c $cpus symphonia_amd/build/pcie_duplex | grep -E "256 workgroups|engine H2D \|\| engine D2H$" | tee -a $OUT/r06i_numa.txt
done
echo '{"node": "all"}' >> $OUT/r06i_decoders.jsonl
$B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct | tee -a $OUT/r06i_decoders.jsonl
