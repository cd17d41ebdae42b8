#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one distillation step (no sync) vs the GPU to run it?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench
from nf_distillation_amd.training import DistillationStep
from param_util import synthetic_images

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
student, teacher = bench.make_models(dev)
student.set_compute_dtype(dtype); teacher.set_compute_dtype(dtype)
step = DistillationStep(student, teacher)
x0 = synthetic_images(64, seed=1)[0].to(dev)
for _ in range(3):
    step.training_step(x0.clone())
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(5):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step.training_step(x0.clone())
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"host enqueue {1e3*(t2-t1):.2f} ms, total {1e3*(t3-t1):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step.training_step(x0.clone())
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
