#!/bin/bash
# tools/prof_trace.sh TAG -- python3 script args…   : rocprofv3 kernel trace of a command on the GPU box (via gpurun);
# prints count / avg / total per kernel and leaves the CSVs under gpurun_out/TAG/
set -o pipefail
TAG=$1; shift; [ "$1" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- "$@" > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")[:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
with open("$O/kernel_summary.txt", "w") as out:
    for k, v in rows[:40]:
        v2 = sorted(v)
        line = "%-100s n=%6d  avg %10.2f us  median %10.2f us  total %12.1f us" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2], sum(v))
        print(line); out.write(line + "\n")
PY
