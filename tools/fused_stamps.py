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
student, teacher = bench.make_models(dev, "bf16")
x = synthetic_images(64, seed=1)[0].to(dev)
lib = capi.lib()
names = ["mix", "im2col", "conv1", "conv2+3 chunks", "tail"]
CASES = []
for ways in (0, 4):
    for nlayers, level in ((2, 1), (35, 2), (69, 3)):
        CASES.append((ways, nlayers, f"level {level}, hidden split {ways}"))
for ways, nlayers, label in CASES:
    e = FlowEngine(list(teacher.flow.layers)[:nlayers])
    e.set_compute_dtype("bf16")
    e.fused_split_inference = ways
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    ld = torch.zeros(64, device=dev)
    with torch.no_grad():
        e.encode(x.clone(), None, ld)
        torch.cuda.synchronize()
        lib.nfk_fused_set_stamps(buf.data_ptr())
        e.encode(x.clone(), None, ld)
        torch.cuda.synchronize()
        lib.nfk_fused_set_stamps(None)
    st = buf.view(-1, 16).cpu()
    st = st[st[:, 0] != 0]
    d = (st[:, 1:6] - st[:, 0:5]).double()
    tot = (st[:, 5] - st[:, 0]).double()
    if (st[:, 6] != 0).any():
        ex = [(st[:, i] - st[:, 3]).double().mean().item() for i in range(6, 16) if (st[:, i] != 0).any()]
        print("   extra stamps (cycles after stamp 3):", " ".join(f"{v:.0f}" for v in ex))
    print(f"{label}: {len(st)} WGs, total {tot.mean():.0f} cycles (max {tot.max():.0f}); " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0).tolist())), flush=True)
