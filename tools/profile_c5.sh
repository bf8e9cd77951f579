#!/bin/bash
# tools/profile_c5.sh TAG — rocprofv3 kernel trace + MFMA counters for the d = 64 config (run on the GPU box via gpurun)
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_c5_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_configs.py c2 c3 c5 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/tools/bench_configs.py c5 > $O/mfma.log 2>&1 || { tail -5 $O/mfma.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/mfma2 -- python3 $R/tools/bench_configs.py c5 > $O/mfma2.log 2>&1 || { tail -5 $O/mfma2.log; }
cd $R
python3 - <<PY
import csv, glob, json, os, collections
O="$O"; tag="$TAG"
out={}
lines=["# rocprofv3 summary: configs C2 / C3 / C5 ("+tag+")",""]
dur=collections.defaultdict(list)
for f in glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Kernel_Name"].split("(")[0].replace("void ","")[:80]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
lines+=["## kernel trace (tools/bench_configs.py c2 c3 c5)","","| kernel | calls | avg us | total ms |","|---|---|---|---|"]
for k,v in sorted(dur.items(), key=lambda kv:-sum(kv[1]))[:14]:
    lines.append(f"| \`{k}\` | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.2f} |")
for sub in ("mfma","mfma2"):
    cnt=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O+f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rule64s" in r["Kernel_Name"]:
                cnt[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
    if cnt:
        lines+=["",f"## PMC on k_rule64s ({sub} pass, per launch median)","","| counter | value |","|---|---|"]
        for c,d in cnt.items():
            v=sorted(d["v"])[len(d["v"])//2]; out[c]=v; lines.append(f"| {c} | {v:.4g} |")
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
    # MFMA_BUSY counts cycles summed over SIMDs; GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md DVFS note)
    util = out["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    lines += ["", f"MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 256 CU x 4 SIMD) = **{util:.3f}**"]
    out["mfma_util"]=util
os.makedirs("gpurun_out/profiles_c5_"+tag, exist_ok=True)
open("gpurun_out/profiles_c5_"+tag+"/"+tag+"_c5_rocprof.md","w").write("\n".join(lines)+"\n")
json.dump(out, open("gpurun_out/profiles_c5_"+tag+"/"+tag+"_c5_rocprof.json","w"), indent=1)
print("\n".join(lines))
PY
grep -h "config" $O/trace.log | head -5
