#!/bin/bash
# lab: shapes of the scalar chain scan (CX_CHAIN_SHAPE) on C2 and the structured family; run from the repo root on the GPU box
set -e
mkdir -p gpurun_out
for s in ${SHAPES:-0 1 2}; do
  echo "== shape $s" >> gpurun_out/ab_chain.log
  CX_CHAIN_SHAPE=$s python tools/bench_configs.py c2 vmp_structured >> gpurun_out/ab_chain.log 2>&1
done
