#!/bin/bash
# tools/profile_c5_pipes.sh — which pipe of the CU the d = 64 rule kernel keeps busy (run on the GPU box via gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c5_pipes
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $O/a -- python3 $R/tools/bench_configs.py c5 > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -- python3 $R/tools/bench_configs.py c5 > $O/b.log 2>&1 || { tail -5 $O/b.log; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/c -- python3 $R/tools/bench_configs.py c5 > $O/c.log 2>&1 || { tail -5 $O/c.log; }
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("a","b","c"):
    cnt=collections.defaultdict(list)
    for f in glob.glob("$O/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rule64s" in r["Kernel_Name"]: cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in cnt.items():
        v=sorted(v); print(sub, c, "median %.4g" % v[len(v)//2], "n", len(v))
PY
