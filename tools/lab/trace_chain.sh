#!/bin/bash
# lab: kernel trace of C2 + the structured family for one CX_CHAIN_SHAPE (argument), summary of the chain kernels on stdout
set -o pipefail
S=${1:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp CX_CHAIN_SHAPE=$S
O=$R/gpurun_out/trace_chain_$S
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_configs.py c2 vmp_structured > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "chain" in k[0] or "rate" in k[0]:
        v.sort()
        print(f"{k[0]:60s} grid {k[1]:>9s} n {len(v):5d} median {v[len(v)//2]:8.1f} us")
PY
