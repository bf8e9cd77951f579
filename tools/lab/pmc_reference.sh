#!/bin/bash
# L2 hits and misses of the cluster launch on the C4 reference-order row (rocprofv3 --pmc, a pass of its own)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_ref_$tag -- python3 $R/tools/bench_configs.py reference:1415:nofp > $R/gpurun_out/pmc_ref_$tag.log 2>&1
  f=$(find $R/gpurun_out/pmc_ref_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "k_ref_cluster" in k or "k_batch" in k:
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
done
