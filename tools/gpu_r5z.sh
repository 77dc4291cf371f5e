#!/bin/bash
# Round 5, the evidence run on the final tree: tools/gpu_round.sh (kernel stats, HBM traffic, the GPU suite, every bench line, the default
# line) + SQ-counter passes of every kernel the default line names + the decoders line.
bash tools/gpu_round.sh r05z
bash tools/gpu_pmc.sh r05z aac mp3 vorbis vorbisf aacjs mp3q alac flac
cp gpurun_out/r05z_*_sq_counters.txt gpurun_out/profiles_r05z/ 2>/dev/null
timeout 900 python bench.py --workload decoders > gpurun_out/profiles_r05z/r05z_decoders.json 2> gpurun_out/r05z_decoders.err
echo "decoders rc=$?"
ls gpurun_out/profiles_r05z | head -60
