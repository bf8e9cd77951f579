#!/bin/bash
# tools/profile_c5scan.sh TAG [K…] — the dim 64 chain-scan sweep (config C5, exact schedule) under rocprofv3 on the GPU box:
# kernel trace (per-launch durations of ONE sweep, in launch order), then the matrix / vector counters of its kernels in
# separate --pmc passes.  Summary -> gpurun_out/profiles_TAG/TAG_c5scan_rocprof.{md,json} (copy into profiles/).
set -o pipefail
TAG=${1:-r04}; shift
KS=${1:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c5scan_$TAG
mkdir -p $O $R/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
ARG=c5scan; [ -n "$KS" ] && ARG=c5scan:$KS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_configs.py $ARG > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
if [ -z "$CX_PROF_TRACE_ONLY" ]; then
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/tools/bench_configs.py $ARG > $O/mfma.log 2>&1 || { tail -5 $O/mfma.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/mfma2 -- python3 $R/tools/bench_configs.py $ARG > $O/mfma2.log 2>&1 || { tail -5 $O/mfma2.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_configs.py $ARG > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_configs.py $ARG > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
fi
cd $R
python3 tools/summarize_c5scan.py --tag $TAG --dir $O --out $R/gpurun_out/profiles_$TAG
