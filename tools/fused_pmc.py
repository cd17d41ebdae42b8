#!/usr/bin/env python3
"""Tiny driver for `rocprofv3 --pmc ...`: a few teacher forwards (fused FlowStep at every CIFAR level); argv[1] = hidden
split of the small grids (4, 2 or 0)."""
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
e = FlowEngine(list(teacher.flow.layers))
e.set_compute_dtype("bf16")
e.fused_split_inference = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ld = torch.zeros(64, device=dev)
with torch.no_grad():
    for _ in range(3):
        e.encode(x.clone(), None, ld)
torch.cuda.synchronize()
