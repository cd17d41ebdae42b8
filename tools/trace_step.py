#!/usr/bin/env python3
"""Summarise one replayed step from a rocprofv3 kernel trace CSV: per queue, the sequence of kernels (run-length encoded) with
their durations -- to spot small launches on the student's chain.  usage: trace_step.py <p_kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = adam[-3] + 1, adam[-2] + 1          # one full step between two Adam launches, late in the run
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
per_q = defaultdict(list)
for r in step:
    per_q[r["Queue_Id"]].append(r)
print(f"step: {len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us wall")
for q, rs in per_q.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    print(f"\n== queue {q}: {len(rs)} launches, {busy:.1f} us of kernel time")
    small = defaultdict(lambda: [0, 0.0])
    for r in rs:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        small[n][0] += 1
        small[n][1] += d
    for n, (c, d) in sorted(small.items(), key=lambda kv: -kv[1][1]):
        print(f"   {n:60s} x{c:4d} {d:9.1f} us")
    gaps = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(rs, rs[1:])) / 1e3
    print(f"   gaps between consecutive launches of this queue: {gaps:.1f} us")
