"""gaps and long kernels in a rocprofv3 kernel trace (csv): python3 stall_trace_sum.py DIR"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
print(len(rows), "kernels")
for i, (s, e, n) in enumerate(rows):
    dur = (e - s) / 1e6
    gap = (s - rows[i - 1][1]) / 1e6 if i else 0.0
    if dur > 5 or (gap > 5 and "k_rule64w" in n and "k_rule64w" in rows[i - 1][2]):
        print(f"#{i} {n}: {dur:.2f} ms, {gap:.2f} ms after the kernel before it ({rows[i - 1][2]})")
