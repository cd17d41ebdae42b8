#!/usr/bin/env python3
"""The two chains of the pipelined step alone on the chip (hipGraph replay, B = 64, bf16): the student's training step
without a teacher, the teacher's forward without a student -- against the pipelined step that runs them side by side."""
import argparse
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
x = synthetic_images(64, seed=5)[0].to(dev)


def timed(fn, n=40, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


student, teacher = bench.make_models(dev, "bf16")
for p in teacher.parameters():
    p.requires_grad_(False)
step = DistillationStep(student, None, nll_weight=1.0, kd_weight=0.0, lr=5e-4, clip_norm=30.0)
run = CapturedStep(step, x, strict=True)
run.prepare()
print(f"student alone (forward + backward + clip + Adam, no hint loss): {timed(lambda: run.training_step(x)):.3f} ms", flush=True)

xs = x.clone()
for split, runs in (("frugal", True), (4, False)):
    eng = teacher.flow.engine()
    eng.fused_split_inference, eng.fused_runs = split, runs
    with torch.no_grad():
        for _ in range(3):
            teacher(xs.clone(), None, keep=[0, 34, 68, 100])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = teacher(xs, None, keep=[0, 34, 68, 100])
    print(f"teacher forward alone (runs={runs}, split={split}): {timed(g.replay):.3f} ms", flush=True)
