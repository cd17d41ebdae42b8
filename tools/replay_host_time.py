#!/usr/bin/env python3
"""How long does the host spend inside hipGraphLaunch for the whole-step graph (vs the GPU time of the step)?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from nf_distillation_amd.training import CapturedStep, DistillationStep  # noqa: E402
from param_util import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
student, teacher = bench.make_models(dev)
student.set_compute_dtype("bf16")
teacher.set_compute_dtype("bf16")
step = DistillationStep(student, teacher, nll_weight=0.9, kd_weight=0.1, lr=5e-4, clip_norm=30.0)
if "--no-overlap" not in sys.argv:
    step.enable_concurrency()
x = synthetic_images(64, seed=1)[0].to(dev)
runner = CapturedStep(step, x)
runner.prepare()
for _ in range(3):
    runner.training_step(x.clone())
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    xs = x.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.training_step(xs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
print(f"host time inside training_step (copy + hipGraphLaunch): {1e3 * sum(host) / len(host):.2f} ms; "
      f"until the GPU is done: {1e3 * sum(total) / len(total):.2f} ms")
