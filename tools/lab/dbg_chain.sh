#!/bin/bash
# lab (timing only, wrong results): CX_CHAIN_DBG bit 16 = the totals kernel loads its links interleaved (coalesced) instead of in runs
for d in 0 16; do
  echo "== dbg $d"
  rm -rf gpurun_out/trace_chain_1
  CX_CHAIN_DBG=$d tools/lab/trace_chain.sh 1 2>&1 | grep -v "true, false\|side" || true
done
