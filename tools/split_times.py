#!/usr/bin/env python3
"""Lone Split2d layers at the CIFAR batch-64 geometries, forward + backward, a few iterations each: run under
`rocprofv3 --kernel-trace --stats` to read the fused kernels' durations without the rest of the step around them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import nf_distillation_amd as nfa  # noqa: E402

for (B, C, H, W) in [(64, 12, 16, 16), (64, 24, 8, 8)]:
    lay = nfa.Split2d(num_channels=C).cuda()
    with torch.no_grad():
        lay.conv.conv.weight.normal_(0, 0.05)
    x = torch.randn(B, C, H, W, device="cuda")
    ld = torch.zeros(B, device="cuda")
    for _ in range(10):
        xx = x.clone().requires_grad_(True)
        z1, ldo = lay(xx, ld.clone().requires_grad_(True))
        (z1.sum() + ldo.sum()).backward()
    torch.cuda.synchronize()
    print(B, C, H, W, "done")
