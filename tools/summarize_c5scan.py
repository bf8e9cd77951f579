#!/usr/bin/env python3
"""Summary of tools/profile_c5scan.sh: the launches of ONE dim 64 chain-scan sweep in launch order (kernel, workgroups, duration),
the totals per kernel, and — when the counter passes ran — matrix-pipe busy fraction, vector and matrix instruction counts and
HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, KiB units, separate passes: MI355X_MICROARCH.md §HBM) per sweep."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

sources_sha16 = import_module("cortex.jl_amd.build").sources_sha16


def short(name):
    return name.split("(")[0].replace("void ", "").replace("cx::", "")[:60]


def rows_of(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True); ap.add_argument("--dir", required=True); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    tr = [r for r in rows_of(os.path.join(a.dir, "trace"), "*kernel_trace.csv") if "compose64" in r["Kernel_Name"] or ("walk64" in r["Kernel_Name"] or "step64" in r["Kernel_Name"])]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one sweep = the launches from one level-0 composition (the largest compose grid) to the next
    grid = lambda r: int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // 64
    big = max((grid(r) for r in tr if "compose64" in r["Kernel_Name"]), default=0)
    starts = [i for i, r in enumerate(tr) if "compose64" in r["Kernel_Name"] and grid(r) == big]
    res = {"tag": a.tag, "sweeps_traced": len(starts), "launches": [], "per_kernel": {}, "counters": {}}
    lines = [f"# rocprofv3 summary: dim 64 chain-scan sweep, config C5 ({a.tag})", "",
             "`tools/profile_c5scan.sh`: `rocprofv3 --kernel-trace --stats` over `python3 tools/bench_configs.py c5scan`; counters in separate `--pmc` passes.", ""]
    if len(starts) >= 2:
        per_sweep = []
        for s0, s1 in zip(starts[:-1], starts[1:]):
            per_sweep.append(tr[s0:s1])
        last = per_sweep[-1]
        n = len(last)
        same = [sw for sw in per_sweep if len(sw) == n]
        lines += [f"Launches of one sweep (median over {len(same)} traced sweeps), in launch order:", "",
                  "| # | kernel | workgroups (waves) | median us | gap to the previous launch us |", "|---|---|---|---|---|"]
        tot = 0.0
        for i in range(n):
            durs = sorted((int(sw[i]["End_Timestamp"]) - int(sw[i]["Start_Timestamp"])) / 1e3 for sw in same)
            gaps = sorted((int(sw[i]["Start_Timestamp"]) - int(sw[i - 1]["End_Timestamp"])) / 1e3 for sw in same) if i else [0.0]
            d, g = durs[len(durs) // 2], gaps[len(gaps) // 2]
            tot += d + g
            res["launches"].append({"kernel": short(last[i]["Kernel_Name"]), "workgroups": grid(last[i]), "median_us": d, "gap_us": g})
            lines.append(f"| {i} | `{short(last[i]['Kernel_Name'])}` | {grid(last[i])} | {d:.1f} | {g:.1f} |")
        wall = sorted((int(sw[-1]["End_Timestamp"]) - int(sw[0]["Start_Timestamp"])) / 1e3 for sw in same)
        res["sweep_us"] = wall[len(wall) // 2]
        lines += ["", f"First launch start to last launch end: **{res['sweep_us'] / 1e3:.3f} ms** per sweep (median).", ""]
    agg = collections.defaultdict(list)
    for r in tr:
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines += ["| kernel | launches | total ms | share |", "|---|---|---|---|"]
    tot_all = sum(sum(v) for v in agg.values()) or 1.0
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        res["per_kernel"][k] = {"launches": len(v), "total_ms": sum(v) / 1e3, "sources_sha16": sources_sha16(k)}
        lines.append(f"| `{k}` | {len(v)} | {sum(v) / 1e3:.2f} | {sum(v) / tot_all:.2f} |")
    # counters: sums per kernel over the whole run divided by the sweeps of that run
    for sub in ("mfma", "mfma2", "fetch", "write"):
        rows = rows_of(os.path.join(a.dir, sub), "*counter_collection.csv")
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(int)
        for r in rows:
            if "compose64" in r["Kernel_Name"] or ("walk64" in r["Kernel_Name"] or "step64" in r["Kernel_Name"]):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(short(r["Kernel_Name"]), r["Counter_Name"])] += 1
        for k, cs in acc.items():
            res["counters"].setdefault(k, {}).update({c: v for c, v in cs.items()})
            res["counters"][k].update({c + "_dispatches": cnt[(k, c)] for c in cs})
    if res["counters"]:
        lines += ["", "Counters, summed over every dispatch of the run (all sweeps incl. warm-up; ratios are what matter):", "",
                  "| kernel | MFMA busy / SQ busy cycles | SQ_INSTS_VALU per SQ_INSTS_MFMA | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | SQ_WAIT_ANY / SQ_WAVE_CYCLES | HBM read GB | HBM write GB |", "|---|---|---|---|---|---|---|"]
        for k, c in res["counters"].items():
            g = lambda n: c.get(n, float("nan"))
            busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES") if g("SQ_BUSY_CYCLES") else float("nan")
            # SQ_BUSY_CYCLES counts per SE-level unit; the ratio MFMA_BUSY / (GRBM_GUI_ACTIVE x SIMDs) is printed beside it
            gui = g("GRBM_GUI_ACTIVE")
            per_simd = g("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) if gui == gui and gui else float("nan")
            res["counters"][k]["mfma_busy_per_simd_cycle"] = per_simd
            lines.append(f"| `{k}` | {busy:.3f} (per SIMD cycle: {per_simd:.3f}) | {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA') if g('SQ_INSTS_MFMA') else float('nan'):.2f} | "
                         f"{g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else float('nan'):.2f} | {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else float('nan'):.2f} | "
                         f"{g('FETCH_SIZE') * 1024 * 2 / 1e9:.2f} | {g('WRITE_SIZE') * 1024 / 1e9:.2f} |")
    open(os.path.join(a.out, f"{a.tag}_c5scan_rocprof.md"), "w").write("\n".join(lines) + "\n")
    json.dump(res, open(os.path.join(a.out, f"{a.tag}_c5scan_rocprof.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
