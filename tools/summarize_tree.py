"""tools/profile_tree.sh's summary: HBM bytes of one exact sweep of the tree schedule = (counters over a process with 24 sweeps - with 4) / 20,
FETCH_SIZE x 1024 x 2 (gfx950: 64 B counted per 128-B request, MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024."""
import argparse
import csv
import glob
import json
import os
import sys
from importlib import import_module

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def total(d, counter):
    s, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                s += float(r["Counter_Value"]); n += 1
    return s, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True); ap.add_argument("--dir", required=True); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    sha = import_module("cortex.jl_amd.build").sources_sha16
    out = {"tag": a.tag, "method": "whole-process FETCH_SIZE / WRITE_SIZE with 24 and with 4 sweeps; (difference) / 20", "n_factors": {"tree": 30000, "tree-deep": 20000}, "rows": {}}
    lines = [f"# HBM traffic of one exact sweep of the tree schedule ({a.tag})", "",
             "`tools/profile_tree.sh`: FETCH_SIZE and WRITE_SIZE in separate passes over `tools/tree_sweeps.py` with 4 and with 24 sweeps; one sweep = (difference) / 20; "
             "HBM bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024.", "", "| forest | dispatches per sweep | HBM read MB | HBM write MB | total MB |", "|---|---|---|---|---|"]
    for shape, row in (("random", "tree"), ("deep", "tree-deep")):
        f4, n4 = total(os.path.join(a.dir, f"{shape}_4_FETCH_SIZE"), "FETCH_SIZE"); f24, n24 = total(os.path.join(a.dir, f"{shape}_24_FETCH_SIZE"), "FETCH_SIZE")
        w4, _ = total(os.path.join(a.dir, f"{shape}_4_WRITE_SIZE"), "WRITE_SIZE"); w24, _ = total(os.path.join(a.dir, f"{shape}_24_WRITE_SIZE"), "WRITE_SIZE")
        rd, wr = (f24 - f4) / 20 * 1024 * 2, (w24 - w4) / 20 * 1024
        out["rows"][row] = {"hbm_read_bytes_per_sweep": rd, "hbm_write_bytes_per_sweep": wr, "hbm_bytes_per_sweep": rd + wr, "dispatches_per_sweep": (n24 - n4) / 20,
                            "sources_sha16": sha("k_batch") + sha("k_chain_")}
        lines.append(f"| {row} | {(n24 - n4) / 20:.0f} | {rd / 1e6:.2f} | {wr / 1e6:.2f} | {(rd + wr) / 1e6:.2f} |")
    os.makedirs(a.out, exist_ok=True)
    json.dump(out, open(os.path.join(a.out, f"{a.tag}_tree_traffic.json"), "w"), indent=1)
    open(os.path.join(a.out, f"{a.tag}_tree_traffic.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
