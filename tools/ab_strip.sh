#!/bin/bash
# tools/ab_strip.sh FLAGS... — A/B of the fused sweep's load / store policies on ONE rank of the 8-way cut of C4 (tools/bench_strip.py, depth 16,
# no exchange) and on the whole grid measured beside it: CX_NT_FLAGS bits 1 scatter stores, 2 message loads, 4 marginal stores nontemporal;
# "auto" = the library's choice by footprint.  One line per variant -> gpurun_out/ab_strip.jsonl (tools/ab_strip_print.py prints the table)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab_strip.jsonl
: > $O
for nt in "${@:-auto}"; do
    if [ "$nt" != auto ]; then export CX_NT_FLAGS=$nt; else unset CX_NT_FLAGS; fi
    echo "{\"variant\": \"nt $nt\"}" >> $O
    timeout -k 10 200 python3 $R/tools/bench_strip.py --depth 16 --sweeps 4000 >> $O 2>> $R/gpurun_out/ab_strip.err || exit 1
done
