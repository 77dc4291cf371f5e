#!/bin/bash
# Round 6: the batcher's three-stage device pipeline without the batcher (tools/ubench/pcie_pipeline.hip)
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 ./build_ab/pcie_pipeline 2>&1 | tee $OUT/r06w_pcie_pipeline.txt
