#!/bin/bash
# tools/profile_c2.sh — run on the GPU box (via gpurun): kernel trace of config C2 (chain scan) and of the variational
# iteration, stats copied to gpurun_out/profiles_c2/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c2
mkdir -p $O $R/gpurun_out/profiles_c2
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/tools/bench_configs.py c2 > $O/c2.log 2>&1 || { tail -5 $O/c2.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vmp -- python3 $R/tools/bench_configs.py vmp > $O/vmp.log 2>&1 || { tail -5 $O/vmp.log; exit 1; }
cd $R
find $O/c2 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/profiles_c2/c2_kernel_stats.csv \;
find $O/vmp -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/profiles_c2/vmp_kernel_stats.csv \;
tail -2 $O/c2.log; tail -3 $O/vmp.log
