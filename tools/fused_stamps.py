#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the fused FlowStep kernel at each CIFAR level
(teacher layers truncated so that the LAST fused launch is at the level of interest)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from nf_distillation_amd import capi  # noqa: E402
from nf_distillation_amd.engine import FlowEngine  # noqa: E402
from param_util import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
student, teacher = bench.make_models(dev)
x = synthetic_images(64, seed=1)[0].to(dev)
lib = capi.lib()
names = ["mix", "im2col", "conv1", "conv2+3 chunks", "tail"]
CASES = []
for gen, base in (("gen2", 0x100), ("gen1", 0x300)):
    for nlayers, level in ((2, 1), (35, 2), (69, 3)):
        CASES.append((base, nlayers, f"{gen} level {level}"))
CASES += [(0x301, 2, "gen1 level 1, no in-loop weight loads"), (0x302, 2, "gen1 level 1, no conv3")]
for dbg, nlayers, label in CASES:
    lib.nfk_fused_set_debug(dbg)
    e = FlowEngine(list(teacher.flow.layers)[:nlayers])
    e.set_compute_dtype("bf16")
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    ld = torch.zeros(64, device=dev)
    with torch.no_grad():
        e.encode(x.clone(), None, ld)
        torch.cuda.synchronize()
        lib.nfk_fused_set_stamps(buf.data_ptr())
        e.encode(x.clone(), None, ld)
        torch.cuda.synchronize()
        lib.nfk_fused_set_stamps(None)
    lib.nfk_fused_set_debug(0)
    st = buf.view(-1, 16).cpu()
    st = st[st[:, 0] != 0]
    d = (st[:, 1:6] - st[:, 0:5]).double()
    tot = (st[:, 5] - st[:, 0]).double()
    if (st[:, 6] != 0).any():
        ex = [(st[:, i] - st[:, 3]).double().mean().item() for i in range(6, 16) if (st[:, i] != 0).any()]
        print("   extra stamps (cycles after stamp 3):", " ".join(f"{v:.0f}" for v in ex))
    print(f"{label}: {len(st)} WGs, total {tot.mean():.0f} cycles (max {tot.max():.0f}); " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0).tolist())), flush=True)
