#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + separate --pmc passes) into profiles/<tag>_*.{md,json}.

    python tools/summarize_rocprof.py --tag r01 --trace DIR --fetch DIR --write DIR [--calib-fetch DIR --calib-write DIR]

Traffic correction follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads
1/2 of a 16 B/lane coalesced stream, other widths are calibrated here on known byte counts (tools/hbm_calib.hip)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def kernel_durations(d):
    out = defaultdict(list)
    for f in find(d, "*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            out[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)  # us
    return out


def counter_per_kernel(d, counter):
    out = defaultdict(list)
    for f in find(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return out


def short(name):
    return name.split("(")[0].replace("void ", "")[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--trace"); ap.add_argument("--fetch"); ap.add_argument("--write")
    ap.add_argument("--calib-fetch"); ap.add_argument("--calib-write")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--kernel", default="k_sweep")
    ap.add_argument("--reads", default="", help="expected read bytes by width, e.g. 16:160e6,8:64e6,4:40e6,1:2e6")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    res = {"tag": a.tag}
    lines = [f"# rocprofv3 summary {a.tag}", ""]
    if a.trace:
        dur = kernel_durations(a.trace)
        tot = sum(sum(v) for v in dur.values())
        lines += ["## kernel trace (--kernel-trace --stats)", "", "| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
        res["kernels"] = {}
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            # drop the first 10% (warmup) only for the average of the dominant kernels
            res["kernels"][short(k)] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v), "total_ms": sum(v) / 1e3}
            lines.append(f"| `{short(k)}` | {len(v)} | {sum(v)/len(v):.2f} | {min(v):.2f} | {max(v):.2f} | {sum(v)/1e3:.3f} | {100*sum(v)/tot:.1f} |")
        lines.append("")
    calib = {}
    if a.calib_fetch and a.calib_write:
        cf, cw = counter_per_kernel(a.calib_fetch, "FETCH_SIZE"), counter_per_kernel(a.calib_write, "WRITE_SIZE")
        known = float(1 << 30)
        lines += ["## counter calibration on known byte counts (tools/hbm_calib.hip, 1 GiB per kernel)", "",
                  "| kernel | counter | counter KiB*1024 | known bytes | bytes per counted byte |", "|---|---|---|---|---|"]
        for k, v in sorted(cf.items()):
            if "calib_read" in k:
                m = sorted(v)[len(v) // 2] * 1024
                calib[short(k)] = known / m
                lines.append(f"| `{short(k)}` | FETCH_SIZE | {m:.4g} | {known:.4g} | {known/m:.3f} |")
        for k, v in sorted(cw.items()):
            if "calib_write" in k:
                m = sorted(v)[len(v) // 2] * 1024
                calib[short(k)] = known / m
                lines.append(f"| `{short(k)}` | WRITE_SIZE | {m:.4g} | {known:.4g} | {known/m:.3f} |")
        lines.append("")
        res["calibration"] = calib
    if a.fetch and a.write:
        f, w = counter_per_kernel(a.fetch, "FETCH_SIZE"), counter_per_kernel(a.write, "WRITE_SIZE")
        lines += ["## HBM traffic per launch (separate --pmc passes)", "",
                  "| kernel | launches | FETCH_SIZE KiB*1024 | WRITE_SIZE KiB*1024 |", "|---|---|---|---|"]
        res["traffic"] = {}
        for k in sorted(set(f) | set(w)):
            fv, wv = f.get(k, [0]), w.get(k, [0])
            fm, wm = sorted(fv)[len(fv) // 2] * 1024, sorted(wv)[len(wv) // 2] * 1024
            res["traffic"][short(k)] = {"fetch_raw_bytes": fm, "write_raw_bytes": wm, "launches": len(fv)}
            lines.append(f"| `{short(k)}` | {len(fv)} | {fm:.4g} | {wm:.4g} |")
        lines.append("")
        dom = [k for k in res["traffic"] if a.kernel in k]
        if dom:
            k = max(dom, key=lambda x: res["traffic"][x]["fetch_raw_bytes"])
            t = res["traffic"][k]
            # guide's gfx950 correction: double FETCH_SIZE for the wide (16 B/lane) coalesced streams; if a width
            # calibration is available, weight the correction by the kernel's expected read mix
            corr = 2.0
            note = "FETCH_SIZE x2 (gfx950, 16 B/lane coalesced reads: MI355X_MICROARCH.md §HBM)"
            if calib and a.reads:
                mix = {int(p.split(":")[0]): float(p.split(":")[1]) for p in a.reads.split(",")}
                cal_by_w = {}
                for name, c in calib.items():
                    if "calib_read<int>" in name: cal_by_w[4] = c
                    if "calib_read<double>" in name: cal_by_w[8] = c
                    if "calib_read<HIP_vector_type" in name or "double2" in name and "read" in name: cal_by_w[16] = c
                cal_by_w.setdefault(1, cal_by_w.get(4, 1.0))
                counted = sum(b / cal_by_w.get(wd, 2.0) for wd, b in mix.items())
                corr = sum(mix.values()) / counted
                note = f"FETCH_SIZE x{corr:.3f}: per-width calibration {cal_by_w} weighted by the kernel's read mix {mix}"
            wcorr = 1.0
            for name, c in calib.items():
                if "calib_write<" in name: wcorr = c
            hbm = t["fetch_raw_bytes"] * corr + t["write_raw_bytes"] * wcorr
            res["dominant"] = {"kernel": k, "hbm_bytes_per_launch": hbm, "fetch_corrected": t["fetch_raw_bytes"] * corr,
                               "write_corrected": t["write_raw_bytes"] * wcorr, "correction": note}
            lines += [f"**dominant kernel** `{k}`: corrected HBM bytes per launch = **{hbm:.4g}** "
                      f"(reads {t['fetch_raw_bytes']*corr:.4g} + writes {t['write_raw_bytes']*wcorr:.4g}); {note}", ""]
            import sys
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from importlib import import_module
            sha = import_module("cortex.jl_amd.build").sources_sha16("k_sweep")
            json.dump({"kernel": "k_sweep<fused>", "hbm_bytes_per_launch": hbm, "updates_per_launch": 16006480, "source": f"profiles/{a.tag}_rocprof.md",
                       "sources_sha16": sha, "sources_note": "sha256[:16] of the kernel's sources (cortex.jl_amd/build.py: sources_sha16): bench.py "
                                                             "refuses this figure once they change"},
                      open(os.path.join(a.out, "traffic_latest.json"), "w"))
    open(os.path.join(a.out, f"{a.tag}_rocprof.md"), "w").write("\n".join(lines) + "\n")
    json.dump(res, open(os.path.join(a.out, f"{a.tag}_rocprof.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
