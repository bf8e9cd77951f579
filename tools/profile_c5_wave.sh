#!/bin/bash
# tools/profile_c5_wave.sh — PMC passes on the d = 64 rule kernel selected by CX_RULE64 (w: wave per message); run on the GPU box via gpurun
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c5_wave
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/tools/bench_configs.py c5 > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -- python3 $R/tools/bench_configs.py c5 > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/tools/bench_configs.py c5 > $O/p3.log 2>&1 || { tail -5 $O/p3.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("p1","p2","p3"):
    cnt=collections.defaultdict(list)
    for f in glob.glob("$O/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rule64" in r["Kernel_Name"]:
                cnt[(r["Kernel_Name"].split("(")[0][-12:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for c,v in sorted(cnt.items()):
        v=sorted(v); print(sub, c, "median %.4g" % v[len(v)//2], "per message %.1f" % (v[len(v)//2]/199998), "n", len(v))
PY
