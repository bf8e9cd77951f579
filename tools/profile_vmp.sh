#!/bin/bash
# tools/profile_vmp.sh TAG — the two variational families (tools/bench_configs.py vmp_structured / vmp_mean_field, n = 1e6
# states, 60 iterations each) under rocprofv3 on the GPU box: kernel trace, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes.
# Summary -> gpurun_out/profiles_TAG/TAG_vmp_rocprof.md (copy into profiles/).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export CX_BENCH_REPS=1      # one batch per row: the summaries count launches per iteration
mkdir -p $R/gpurun_out/profiles_$TAG
for fam in structured mean_field; do
  O=$R/gpurun_out/prof_vmp_${fam}_$TAG
  mkdir -p $O
  cd /tmp
  B="python3 $R/tools/bench_configs.py vmp_$fam"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
  cd $R
done
R=$R TAG=$TAG python3 - <<'PY'
import csv, glob, collections, json, os
R = os.environ["R"]; TAG = os.environ["TAG"]
import sys
sys.path.insert(0, R)
from importlib import import_module
_sha = import_module("cortex.jl_amd.build").sources_sha16
SHA = _sha("k_mf_normal") + _sha("k_chain_apply")       # the families' own kernels + the chain scan the structured one runs
ITER = 60      # bench_configs.vmp: 10 warm-up + 50 timed iterations
def per_kernel(d, counter=None):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv" if counter else "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
            if counter:
                if r["Counter_Name"] == counter: out[k].append(float(r["Counter_Value"]))
            else:
                out[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out
lines = ["# rocprofv3 summary: variational families (SURVEY §8 f3), n = 1e6 states, 5,999,997 edges (" + TAG + ")", "",
         "`tools/profile_vmp.sh`: kernel trace, FETCH_SIZE and WRITE_SIZE in separate passes (HBM bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024), "
         "60 iterations (all latent states, then both precisions) per family.  Per-iteration figures count the kernels launched at least once per iteration.", ""]
res = {}
for fam in ("structured", "mean_field"):
    O = f"{R}/gpurun_out/prof_vmp_{fam}_{TAG}"
    dur, fe, wr = per_kernel(O + "/trace"), per_kernel(O + "/fetch", "FETCH_SIZE"), per_kernel(O + "/write", "WRITE_SIZE")
    lines += [f"## {fam}", "", "| kernel | launches per iteration | avg us | HBM read MB / launch | HBM write MB / launch |", "|---|---|---|---|---|"]
    t_us = b_tot = 0.0
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        per = len(v) / ITER
        if per < 0.9: continue
        v2 = sorted(v); avg = sum(v2[: max(1, len(v2) - 2)]) / max(1, len(v2) - 2)
        rd = (sorted(fe[k])[len(fe[k]) // 2] * 2048) if k in fe else 0.0
        ww = (sorted(wr[k])[len(wr[k]) // 2] * 1024) if k in wr else 0.0
        n = round(per)
        t_us += n * avg; b_tot += n * (rd + ww)
        lines.append(f"| `{k}` | {n} | {avg:.1f} | {rd/1e6:.2f} | {ww/1e6:.2f} |")
    t_us = t_us or float("nan")
    res[fam] = {"kernel_us_per_iteration": t_us, "hbm_bytes_per_iteration": b_tot, "GBps": b_tot / t_us / 1e3,
                "sources_sha16": SHA, "sources_note": "sha256[:16] over cx_vmp.hip and the chain-scan kernels' sources: the figure is refused once they change"}
    lines += ["", f"per iteration: {t_us:.1f} us of kernels, {b_tot/1e6:.1f} MB of HBM traffic = {b_tot/t_us/1e3:.0f} GB/s = **{b_tot/t_us/1e3/8000:.2f}** of the 8 TB/s peak", ""]
open(f"{R}/gpurun_out/profiles_{TAG}/{TAG}_vmp_rocprof.md", "w").write("\n".join(lines) + "\n")
json.dump(res, open(f"{R}/gpurun_out/profiles_{TAG}/{TAG}_vmp_rocprof.json", "w"), indent=1)
print("\n".join(lines))
PY
