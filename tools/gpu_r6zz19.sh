#!/bin/bash
# Round 6 (last session): the ILP scheduling strategy for vorbis_wg.hip (8192-sample blocks; 96 B of scratch under it) on the big block-size pairs
bash tools/gpu_pairs_ab.sh r06zz19 "10,13 12,13 11,13" 2 symphonia_amd/libsymaccel.so build_ab/ilp_wg.so
