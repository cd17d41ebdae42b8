#!/usr/bin/env python3
"""Per-queue kernel sequences of ONE replayed step, from a rocprofv3 --kernel-trace CSV of bench.py
(usage: step_timeline.py <dir with *_kernel_trace.csv> [max lines per queue])."""
import collections
import csv
import glob
import itertools
import sys

t = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(t)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
adam = [e for e in ev if 'adam_kernel' in e[2]]
a0, a1 = adam[-3][1], adam[-2][1]
step = [e for e in ev if e[0] >= a0 and e[1] <= a1 + 1]
busy, cs, ce = 0, None, None
for s, e, _, _ in step:
    if ce is None or s > ce:
        busy += (ce - cs) if ce is not None else 0
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"step wall {(a1 - a0) / 1e3:.0f} us, {len(step)} kernels, some kernel running {busy / 1e3:.0f} us")


def short(n):
    return n.replace('void nfk::', '').replace('nfk::', '').split('(')[0][:40]


for q, cnt in collections.Counter(e[3] for e in step).most_common():
    es = [e for e in step if e[3] == q]
    print(f"== queue {q}: {cnt} kernels, {sum(e[1] - e[0] for e in es) / 1e3:.0f} us of kernel time, "
          f"first {(es[0][0] - a0) / 1e3:.0f} last {(es[-1][1] - a0) / 1e3:.0f}")
    out = []
    for k, g in itertools.groupby(es, key=lambda e: short(e[2])):
        g = list(g)
        out.append(f"{(g[0][0] - a0) / 1e3:7.0f}us {k} x{len(g)} ({sum(x[1] - x[0] for x in g) / 1e3:.0f}us)")
    print("\n".join(out[:lim]))
