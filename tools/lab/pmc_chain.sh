#!/bin/bash
# lab: SQ counters of the scalar chain-scan kernels on the structured family (CX_CHAIN_SHAPE = $1); two --pmc passes, no trace domains
set -o pipefail
S=${1:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp CX_CHAIN_SHAPE=$S
O=$R/gpurun_out/pmc_chain_$S
mkdir -p $O
cd /tmp
B="python3 $R/tools/bench_configs.py vmp_structured"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    if "chain" not in k and "rate" not in k: continue
    print(k)
    for n, v in sorted(c.items()):
        v.sort()
        print(f"    {n:28s} {v[len(v)//2]:16.0f}")
PY
