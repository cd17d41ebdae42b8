#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the level-mode backward kernels (student K=8, B=64), one level at a time:
the stamps are those of the LAST step each launch walks.  usage: python tools/level_stamps.py [multi=1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from nf_distillation_amd import capi  # noqa: E402
from nf_distillation_amd.engine import FlowEngine  # noqa: E402
from param_util import synthetic_images  # noqa: E402

multi = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
dev = torch.device("cuda", 0)
student, teacher = bench.make_models(dev, "bf16")
lib = capi.lib()
names = ["stage", "im2col", "dgrad3+epi", "save g_a2", "dgrad2 k-loop", "w1 ring + barrier", "epilogue2", "prefetch + save g_a1",
         "dgrad1", "(d1 add) g_x", "mix partials", "barrier", "coupling", "finish"]
layers = list(student.flow.layers)
shapes = {1: (3, 32, 32), 2: (6, 16, 16), 3: (12, 8, 8)}
for level, (lo, hi) in ((1, (0, 9)), (2, (10, 19)), (3, (20, 29))):
    sub = layers[lo:hi]   # squeeze + 8 steps (no split: the level alone)
    e = FlowEngine(sub)
    e.set_compute_dtype("bf16")
    e.level_multi = multi
    x = torch.randn((64,) + shapes[level], device=dev) * 0.3
    buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    ld = torch.zeros(64, device=dev)
    for rep in range(2):
        outs, ldo = e.encode(x.clone().requires_grad_(True), None, ld)
        if rep == 1:
            torch.cuda.synchronize()
            lib.nfk_fused_bwd_set_stamps(buf.data_ptr())
        (outs[-1].square().mean() + ldo.mean()).backward()
        torch.cuda.synchronize()
        lib.nfk_fused_bwd_set_stamps(None)
    st = buf.view(-1, 16).cpu()
    st = st[st[:, 0] != 0]
    d = (st[:, 1:15] - st[:, 0:14]).double()
    tot = (st[:, 14] - st[:, 0]).double()
    print(f"level {level} ({'runs' if multi else 'per step'}): {len(st)} WGs, last step: total {tot.mean():.0f} cycles (max {tot.max():.0f}); " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0).tolist())), flush=True)
