#!/bin/bash
# effective clock and MFMA-pipe occupancy of the bare f64 MFMA loop (tools/lab/mfma64_peak)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_peak
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc -- $R/tools/lab/mfma64_peak > $O/pmc.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $R/tools/lab/mfma64_peak > $O/trace.log 2>&1
python3 - <<PY
import csv, glob, collections
pm = collections.defaultdict(dict)
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        pm[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); pm[int(r["Dispatch_Id"])]["k"] = r["Kernel_Name"]
dur = {}
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for d in sorted(pm):
    p = pm[d]
    t = dur.get(d)
    if not t: continue
    clk = p["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
    print(f"dispatch {d} {p['k'][:40]}: {t*1e3:.3f} ms, clock {clk:.2f} GHz, MFMA busy {p['SQ_VALU_MFMA_BUSY_CYCLES'] / (p['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}, "
          f"{p['SQ_INSTS_MFMA'] * 2048 / t / 1e12:.1f} TFLOP/s")
PY
