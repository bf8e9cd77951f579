#!/bin/bash
# rocprofv3 --kernel-trace --stats over the C4 reference-order row (tools/bench_configs.py reference:1415:nofp): the cluster launch's
# average duration beside the wall clock per call that the row reports
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_reference -- python3 $R/tools/bench_configs.py reference:1415:nofp > $R/gpurun_out/prof_reference.json 2> $R/gpurun_out/prof_reference.err
f=$(find $R/gpurun_out/prof_reference -name "*kernel_stats.csv" | head -1)
head -8 "$f" > $R/gpurun_out/r05_reference_kernel_stats.csv
python3 -c "
import json; r=json.load(open('$R/gpurun_out/prof_reference.json')); print('ms_per_call (wall clock, under the profiler):', r['ms_per_call'], r['plan'])" >> $R/gpurun_out/r05_reference_kernel_stats.csv
cat $R/gpurun_out/r05_reference_kernel_stats.csv
