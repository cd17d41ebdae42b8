#!/usr/bin/env python3
"""One replayed step from a rocprofv3 kernel trace CSV as a full timeline: start (us after the previous Adam), queue, duration,
short kernel name, workgroups.  usage: trace_timeline.py <p_kernel_trace.csv> [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = adam[-3] + 1, adam[-2] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
qs = sorted({r["Queue_Id"] for r in step})
print(f"{len(step)} launches, wall {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, queues {qs}")
for r in step[: int(sys.argv[2]) if len(sys.argv) > 2 else None]:
    n = r["Kernel_Name"].replace("void ", "").replace("nfk::", "")
    n = n.split("(")[0][:58]
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
    wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)
    col = qs.index(r["Queue_Id"])
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} {"        " * col}q{col} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f}us wg={gx // max(wx, 1):5d} {n}')
