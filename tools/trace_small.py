#!/usr/bin/env python3
"""One replayed step from a rocprofv3 kernel trace CSV as a timeline of everything that is NOT a fused FlowStep / merged
weight-gradient launch: start (us after the previous Adam), queue, duration, kernel -- the small launches at the head and tail of
the two chains.  usage: trace_small.py <p_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = adam[-3] + 1, adam[-2] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    n = r["Kernel_Name"]
    if "nfk::fused" in n or "wgrad_tail" in n: continue
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} q{r["Queue_Id"]} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}  {n[:150]}')
