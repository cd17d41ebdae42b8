#!/usr/bin/env python3
"""Micro-benchmark of the bf16 conv GEMM through nfk_step2d_fwd's building blocks is not exposed op-wise, so this
drives whole FlowSteps (five-kernel path) and fused steps at each CIFAR level and reports per-kernel times from
HIP events around individual launches (nfk_profile_*), plus the fused step's wall time."""
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
layers = list(teacher.flow.layers)
# teacher: [squeeze, 32 steps, split, squeeze, 32 steps, split, squeeze, 32 steps]
spans = {"level 1": (0, 33), "level 2": (0, 67), "level 3": (0, 101)}
prev = 0.0
for label, fused, ways in (("5-kernel", False, 0), ("fused, no hidden split", True, 0), ("fused, 4-way hidden split", True, 4)):
    prev_t = 0.0
    for name, (a, b) in spans.items():
        e = FlowEngine(layers[a:b])
        e.set_compute_dtype("bf16")
        e.fused = fused
        e.fused_split_inference = ways
        ld = torch.zeros(64, device=dev)
        with torch.no_grad():
            for _ in range(3):
                e.encode(x.clone(), None, ld)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                e.encode(x.clone(), None, ld)
            t1.record()
            torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        print(f"{label} {name}: cumulative {ms:.3f} ms, this level {ms - prev_t:.3f} ms "
              f"= {(ms - prev_t) / 32 * 1e3:.1f} us/step", flush=True)
        prev_t = ms
