#!/usr/bin/env python3
"""Per-level time of the teacher's (K = 32, B = 64) fused forward: one launch per step (hidden split 0 / 4) against one
launch per level (nfk_run2d_fwd_fused), with the shader-clock stamps of the last step's phases."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from nf_distillation_amd import capi  # noqa: E402
from nf_distillation_amd.engine import FlowEngine  # noqa: E402

dev = torch.device("cuda", 0)
student, teacher = bench.make_models(dev, "bf16")
lib = capi.lib()
for p in teacher.parameters():
    p.requires_grad_(False)
layers = list(teacher.flow.layers)   # [squeeze, 32 steps, split, squeeze, 32 steps, split, squeeze, 32 steps]
LEVELS = {"level 1": (1, 33, (12, 16, 16)), "level 2": (35, 67, (24, 8, 8)), "level 3": (69, 101, (48, 4, 4))}
NAMES = ["mix", "im2col", "conv1", "conv2 loop", "conv2 epilogue", "conv3", "tail"]
for label, (a, b, shape) in LEVELS.items():
    x = torch.randn(64, *shape, device=dev).contiguous(memory_format=torch.channels_last)
    for mode, runs, ways in (("per step, no split", False, 0), ("per step, 4-way split", False, 4), ("one launch per level", True, 0)):
        if label == "level 1" and mode != "per step, no split":
            continue
        e = FlowEngine(layers[a:b])
        e.set_compute_dtype("bf16")
        e.fused_runs, e.fused_split_inference = runs, ways
        ld = torch.zeros(64, device=dev)
        with torch.no_grad():
            for _ in range(3):
                e.encode(x, None, ld)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                e.encode(x, None, ld)
            t1.record()
            torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 20
            buf = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
            lib.nfk_fused_set_stamps(buf.data_ptr())
            e.encode(x, None, ld)
            torch.cuda.synchronize()
            lib.nfk_fused_set_stamps(None)
        st = buf.view(-1, 16).cpu()
        st = st[st[:, 0] != 0]
        order = [0, 1, 2, 3, 6, 7, 4, 5]   # stamp slots in time order
        seq = st[:, order].double()
        d = (seq[:, 1:] - seq[:, :-1]).mean(0).tolist()
        tot = (st[:, 5] - st[:, 0]).double().mean().item()
        ex = {i: (st[:, i] - st[:, 0]).double().mean().item() for i in (8, 9, 10, 1)}
        tl = {i: (st[:, i] - st[:, 4]).double().mean().item() for i in (11, 5)}
        print("   mix phase, cycles after stamp 0: loads issued %.0f, past 1st barrier %.0f, mix loop done %.0f, past 2nd barrier %.0f;"
              " tail after stamp 4: loops done %.0f, end %.0f" % (ex[8], ex[9], ex[10], ex[1], tl[11], tl[5]))
        print(f"{label}, {mode}: {ms * 1e3:.0f} us per level = {ms * 1e3 / 32:.1f} us/step (host clock, eager); last step "
              f"{len(st)} WGs {tot:.0f} cycles: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, d)), flush=True)
