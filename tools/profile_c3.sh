#!/bin/bash
# tools/profile_c3.sh — PMC passes on the d = 4 sweep kernel (config C3); run on the GPU box via gpurun
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c3
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/bench_configs.py c3 > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -- python3 $R/tools/bench_configs.py c3 > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/tools/bench_configs.py c3 > $O/p3.log 2>&1 || { tail -5 $O/p3.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("p1","p2","p3"):
    cnt=collections.defaultdict(list)
    for f in glob.glob("$O/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sweep_mv" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) >= 0:
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in cnt.items():
        v=sorted(v); print(sub, c, "median %.4g" % v[len(v)//2], "n", len(v))
PY
