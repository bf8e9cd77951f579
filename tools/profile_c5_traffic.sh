#!/bin/bash
# tools/profile_c5_traffic.sh — HBM traffic counters of the d = 64 rule kernel (run on the GPU box via gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c5_traffic
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/bench_configs.py c5 > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/bench_configs.py c5 > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("f","w"):
    cnt=collections.defaultdict(list)
    for f in glob.glob("$O/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rule64" in r["Kernel_Name"]: cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in cnt.items():
        v=sorted(v); print(sub, c, "median %.4g (KB units -> GB: %.2f; x2 for 16B-lane reads per the guide)" % (v[len(v)//2], v[len(v)//2]/1e6), "n", len(v))
PY
