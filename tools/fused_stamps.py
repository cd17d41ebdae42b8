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
for dbg, nlayers, label in ((0, 2, "level 1"), (1, 2, "level 1, no in-loop weight loads"), (2, 2, "level 1, no conv3"), (3, 2, "level 1, neither")):
    lib.nfk_fused_set_debug(dbg)
    e = FlowEngine(list(teacher.flow.layers)[:nlayers])
    e.set_compute_dtype("bf16")
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    ld = torch.zeros(64, device=dev)
    with torch.no_grad():
        e.encode(x.clone(), None, ld)
        torch.cuda.synchronize()
        lib.nfk_fused_set_stamps(buf.data_ptr())
        e.encode(x.clone(), None, ld)
        torch.cuda.synchronize()
        lib.nfk_fused_set_stamps(None)
    lib.nfk_fused_set_debug(0)
    st = buf.view(-1, 8).cpu()
    st = st[st[:, 0] != 0]
    d = (st[:, 1:6] - st[:, 0:5]).double()
    tot = (st[:, 5] - st[:, 0]).double()
    print(f"{label}: {len(st)} WGs, total {tot.mean():.0f} cycles (max {tot.max():.0f}); " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0).tolist())), flush=True)
