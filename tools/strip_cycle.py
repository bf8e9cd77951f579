"""tools/strip_cycle.py TRACE_DIR DEPTH — median duration of every kernel of one exchange cycle (exchange kernels + DEPTH sweeps) from a
rocprofv3 kernel trace of tools/bench_strip.py (tools/prof_trace.sh); second half of the run only."""
import collections
import csv
import glob
import statistics as st
import sys


def main():
    d, depth = sys.argv[1], int(sys.argv[2])
    import os
    f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]
    seq = []
    for r in rows:
        n = r["Kernel_Name"]
        k = ("push" if "k_ipc_push" in n or "k_ipc_exchange" in n else "unpack" if "k_ipc_unpack" in n else "sweep" if "k_sweep" in n else "pack" if "k_gather" in n
             else "unpack" if "k_scatter" in n else "rccl" if "ccl" in n.lower() else n[:30])
        seq.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    cycles, cur = [], []
    for s in seq:
        if s[0] in ("push", "pack") and cur:
            cycles.append(cur)
            cur = []
        cur.append(s)
    good = [c for c in cycles if sum(1 for s in c if s[0] == "sweep") == depth and c[0][0] in ("push", "pack")]
    if not good:
        print("no complete cycle found")
        return
    L = collections.Counter(len(c) for c in good).most_common(1)[0][0]
    good = [c for c in good if len(c) == L]
    tot = []
    for j in range(L):
        print("%-8s dur %6.2f us   start-to-next-start %6.2f us" % (good[0][j][0], st.median((c[j][2] - c[j][1]) / 1e3 for c in good),
              st.median(((c[j + 1][1] if j + 1 < L else c[j][2]) - c[j][1]) / 1e3 for c in good)))
    span = [(c[-1][2] - c[0][1]) / 1e3 for c in good]
    print("cycle (first kernel start to last kernel end): median %.2f us = %.2f us per sweep over %d cycles" % (st.median(span), st.median(span) / depth, len(good)))


if __name__ == "__main__":
    main()
