#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE)
into per-kernel MFMA-busy fractions.
usage: pmc_mfma.py <dir of the pass> <out.json> "<command profiled>"
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
reports the sum over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back) -- the gfx94x `MfmaUtil` formula written out, because
ROCm 7.2 ships no gfx950 derived-counter section.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16,
16 per 16x16x32), SQ_WAVE_CYCLES / SQ_BUSY_CYCLES quad-cycles (same guide, cycle-constants table)."""
import collections
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k][r['Counter_Name']] += 1
out = {"how": "one pass, `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES "
              "GRBM_GUI_ACTIVE` of `" + sys.argv[3] + "`; per-launch averages over all launches of the symbol; "
              "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * 256 * GRBM_GUI_ACTIVE / 8)",
       "kernels": {}}
rows = []
for k, c in acc.items():
    n = max(cnt[k].values())
    if c.get("SQ_INSTS_MFMA", 0.0) == 0.0 or c.get("GRBM_GUI_ACTIVE", 0.0) == 0.0:
        continue
    per = {name: v / cnt[k][name] for name, v in c.items()}
    cyc = per["GRBM_GUI_ACTIVE"] / 8.0
    per_launch = {name: round(v, 1) for name, v in per.items()}
    per_launch["launches_sampled"] = n
    per_launch["kernel_cycles"] = round(cyc, 1)
    per_launch["mfma_busy"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc), 4)
    rows.append((per["SQ_VALU_MFMA_BUSY_CYCLES"] * n, k, per_launch))
for _, k, v in sorted(rows, reverse=True):
    out["kernels"][k] = v
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, v in list(out["kernels"].items())[:16]:
    print(f"{k[:80]:80s} x{v['launches_sampled']:4d} cycles {v['kernel_cycles']:10.0f}  mfma_busy {100 * v['mfma_busy']:5.1f} %")
