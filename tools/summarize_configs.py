#!/usr/bin/env python3
"""Summary of tools/profile_configs.sh: per kernel the average duration (kernel trace) and the HBM traffic per launch
(FETCH_SIZE x 2 + WRITE_SIZE, KiB units, separate passes: MI355X_MICROARCH.md §HBM), plus the MFMA counters of k_rule64s."""
import argparse
import collections
import csv
import glob
import json
import os


import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
sources_sha16 = import_module("cortex.jl_amd.build").sources_sha16      # the figure is refused once the kernel's sources change


def short(name):
    return name.split("(")[0].replace("void ", "")[:90]


def per_kernel(d, counter=None):
    out = collections.defaultdict(list)
    pat = "*counter_collection.csv" if counter else "*kernel_trace.csv"
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        for r in csv.DictReader(open(f)):
            if counter:
                if r.get("Counter_Name") == counter:
                    out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            else:
                out[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True); ap.add_argument("--dir", required=True); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    dur = per_kernel(os.path.join(a.dir, "trace"))
    fetch = per_kernel(os.path.join(a.dir, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(a.dir, "write"), "WRITE_SIZE")
    res = {"tag": a.tag, "kernels": {}, "traffic": {}}
    lines = [f"# rocprofv3 summary: configs C2 / C3 / C5 ({a.tag})", "",
             "`tools/profile_configs.sh`: `rocprofv3 --kernel-trace --stats`, then `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes, "
             "each over `python3 tools/bench_configs.py c2 c3 c3scan c5 tilesn:16`.  HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950 counts 64 B per 128-B request: "
             "MI355X_MICROARCH.md §HBM; x2.000 measured for 4/8/16 B-per-lane streams, profiles/r01_rocprof.md) + WRITE_SIZE x 1024, "
             "at the L2-fabric side (Infinity-Cache hits included).", "",
             "| kernel | calls | avg us | median us | HBM read MB / launch | HBM write MB / launch | GB/s at the median |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:20]:
        res["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "median_us": med(v)}
        rd = med(fetch[k]) * 1024 * 2 if k in fetch else None
        wr = med(write[k]) * 1024 if k in write else None
        if rd is not None and wr is not None:
            res["traffic"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "sources_sha16": sources_sha16(k)}
            lines.append(f"| `{k}` | {len(v)} | {sum(v)/len(v):.1f} | {med(v):.1f} | {rd/1e6:.2f} | {wr/1e6:.2f} | {(rd+wr)/med(v)/1e3:.0f} |")
        else:
            lines.append(f"| `{k}` | {len(v)} | {sum(v)/len(v):.1f} | {med(v):.1f} | - | - | - |")
    pm = {}
    for sub, names in (("mfma", ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE")),
                       ("mfma2", ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"))):
        for c in names:
            v = [x for k, vs in per_kernel(os.path.join(a.dir, sub), c).items() if "k_rule64" in k for x in vs]
            if v:
                pm[c] = med(v)
    if pm:
        lines += ["", "## PMC on the d = 64 rule kernel (per launch, median)", "", "| counter | value |", "|---|---|"]
        lines += [f"| {c} | {v:.4g} |" for c, v in pm.items()]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "GRBM_GUI_ACTIVE" in pm:
            util = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (pm["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
            pm["mfma_util"] = util
            lines += ["", f"MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CU x 4 SIMD) = **{util:.3f}**"]
        k64 = [k for k in dur if "k_rule64w<2, 4>" in k] or [k for k in dur if "k_rule64" in k]      # (C5's: the d = 16 row runs k_rule64w<4, 1>)
        if "SQ_INSTS_MFMA" in pm and k64:
            t = med(dur[k64[0]]) * 1e-6
            tf = pm["SQ_INSTS_MFMA"] * 2048 / t / 1e12
            pm["mfma_TFLOPs"] = tf
            lines += [f"SQ_INSTS_MFMA x 2048 flop / median launch = **{tf:.1f} TFLOP/s** = {tf/78.6:.3f} of the 78.6 TFLOP/s f64 matrix peak (spec), "
                      f"{tf/77.3:.3f} of the 77.3 TFLOP/s a bare MFMA loop sustains on this chip (profiles/r02_f64_mfma_sustained.txt)"]
        res["pmc_rule64"] = pm
    open(os.path.join(a.out, f"{a.tag}_configs_rocprof.md"), "w").write("\n".join(lines) + "\n")
    json.dump(res, open(os.path.join(a.out, f"{a.tag}_configs_rocprof.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
