#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV directory and prints, for the LAST few light-tracer calls (a call = everything from one k_pt_keys or the
first kernel after a > 200 us idle gap up to the next such point), every kernel with its start offset, duration and the idle gap in front of it."""
import csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# split into calls at the tracing kernels: a call ends with the last kernel before the next k_pt_keys / big gap
calls = []; cur = []
for i, (s, e, n) in enumerate(rows):
    gap = s - rows[i - 1][1] if i else 0
    if cur and (gap > 100000):
        calls.append(cur); cur = []
    cur.append((s, e, n, gap))
if cur:
    calls.append(cur)
def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n[:60]
want = [c for c in calls if any("k_ptracer" in k[2] for k in c)]
print("%d idle-separated groups, %d with a light-tracer kernel; the last 4:" % (len(calls), len(want)))
for c in want[-4:]:
    t0 = c[0][0]; busy = sum(e - s for s, e, _, _ in c); span = c[-1][1] - t0
    print("\n  group of %d kernels: span %.3f ms, kernels %.3f ms, idle inside %.3f ms" % (len(c), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    for s, e, n, gap in c:
        print("    +%8.1f us  %-46s %8.1f us   (idle before: %6.1f us)" % ((s - t0) / 1e3, short(n)[:46], (e - s) / 1e3, gap / 1e3 if s != t0 else 0.0))
