#!/bin/bash
# tools/profile_round.sh TAG — run on the GPU box (via gpurun): kernel trace + separate PMC passes + calibration,
# raw output under gpurun_out/prof_TAG/, summary written by tools/summarize_rocprof.py into profiles/.
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH --steps 200 --warmup 20 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $BENCH --steps 20 --warmup 3 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $BENCH --steps 20 --warmup 3 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfetch -- $R/tools/hbm_calib > $O/cfetch.log 2>&1 || { tail -5 $O/cfetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cwrite -- $R/tools/hbm_calib > $O/cwrite.log 2>&1 || { tail -5 $O/cwrite.log; exit 1; }
cd $R
python3 tools/summarize_rocprof.py --tag $TAG --trace $O/trace --fetch $O/fetch --write $O/write --calib-fetch $O/cfetch --calib-write $O/cwrite \
   --reads 16:160087440,8:32012960,2:20010930,1:2002225 --out $R/gpurun_out/profiles_$TAG
# keep the small CSV stats next to the summary
find $O/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/profiles_$TAG/${TAG}_kernel_stats.csv \;
