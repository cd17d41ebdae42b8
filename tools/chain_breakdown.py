#!/usr/bin/env python3
"""Kernel-time breakdown of the student's chain (the queue that holds the backward) in ONE replayed pipelined step,
from a rocprofv3 --kernel-trace CSV of bench.py (usage: chain_breakdown.py <dir>).  Durations are valid even though the
profiler serialises the queues."""
import collections
import csv
import glob
import sys

t = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(t)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
adam = [e for e in ev if 'adam_kernel' in e[2]]
a0, a1 = adam[-3][1], adam[-2][1]
q = [e for e in ev if 'prior_bwd' in e[2] and a0 <= e[0] <= a1][0][3]   # the queue that runs the backward
step = [e for e in ev if e[0] >= a0 and e[1] <= a1 + 1 and e[3] == q]
other = [e for e in ev if e[0] >= a0 and e[1] <= a1 + 1 and e[3] != q]


def short(n):
    return n.replace('void nfk::', '').replace('nfk::', '').split('(')[0][:56]


agg = collections.OrderedDict()
for e in step:
    k = short(e[2])
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += (e[1] - e[0]) / 1e3
print(f"student chain: {len(step)} kernels, {sum(v[1] for v in agg.values()):.0f} us of kernel time; "
      f"other queues: {len(other)} kernels, {sum(e[1] - e[0] for e in other) / 1e3:.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{k:58s} x{v[0]:3d} {v[1]:8.1f} us  avg {v[1] / v[0]:6.1f}")
