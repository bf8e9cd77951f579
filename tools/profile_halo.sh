#!/bin/bash
# tools/profile_halo.sh — run on the GPU box (via gpurun): kernel timeline of the partitioned sweep with rank 0 as its own
# RCCL neighbour (bench.py --self-halo); raw trace under gpurun_out/prof_halo/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_halo
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --self-halo --steps 60 --warmup 20 --event-stride 1000000 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep '"metric"' $O/run.log | cut -c1-300
