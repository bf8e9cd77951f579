#!/bin/bash
# tools/profile_configs.sh TAG — configs C2 / C3 / C5 under rocprofv3 on the GPU box (via gpurun): kernel trace, then SEPARATE
# --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md §HBM: FETCH_SIZE x2 on gfx950), then the MFMA counters of
# the d = 64 kernel.  Summary -> gpurun_out/profiles_TAG/TAG_configs_rocprof.{md,json} (copy into profiles/).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_configs_$TAG
mkdir -p $O
export TMPDIR=/tmp
export CX_BENCH_REPS=1      # one batch per row: the summaries count launches per iteration
cd /tmp
B="python3 $R/tools/bench_configs.py c2 c3 c3scan c5 tilesn:16"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/tools/bench_configs.py c5 > $O/mfma.log 2>&1 || { tail -5 $O/mfma.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/mfma2 -- python3 $R/tools/bench_configs.py c5 > $O/mfma2.log 2>&1 || { tail -5 $O/mfma2.log; exit 1; }
cd $R
python3 tools/summarize_configs.py --tag $TAG --dir $O --out $R/gpurun_out/profiles_$TAG
