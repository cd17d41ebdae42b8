#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the fused FlowStep backward at each CIFAR level (student K=8, B=64)."""
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
names = ["stage g_a3 / masks", "im2col", "dgrad3+epi", "save g_a2", "dgrad2 k-loop", "w1 ring + barrier", "epilogue2",
         "save g_a1", "dgrad1", "reduce+store"]
for nlayers, level in ((2, 1),):   # (every step writes the same stamp buffer: the level-1 step, last in backward order, wins)
    layers = list(student.flow.layers)[:nlayers]
    e = FlowEngine(layers)
    e.set_compute_dtype("bf16")
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    ld = torch.zeros(64, device=dev)
    for rep in range(2):
        outs, ldo = e.encode(x.clone(), None, ld)
        if rep == 1:
            torch.cuda.synchronize()
            lib.nfk_fused_bwd_set_stamps(buf.data_ptr())
        (outs[-1].square().mean() + ldo.mean()).backward()
        torch.cuda.synchronize()
        lib.nfk_fused_bwd_set_stamps(None)
    st = buf.view(-1, 16).cpu()
    st = st[st[:, 0] != 0]
    d = (st[:, 1:11] - st[:, 0:10]).double()
    tot = (st[:, 10] - st[:, 0]).double()
    print(f"level {level}: {len(st)} WGs (first step in backward order overwrites the others), total {tot.mean():.0f} cycles "
          f"(max {tot.max():.0f}); " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0).tolist())), flush=True)
