#!/usr/bin/env python3
"""Diagnostic: time of the fused backward's weight-gradient launch (triple_wgrad_bf16_kernel) per CIFAR level with parts of
it left out (nfk_debug_triple_parts), from the HIP-event profile of eager student steps."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from nf_distillation_amd import capi  # noqa: E402
from param_util import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
student, teacher = bench.make_models(dev, "bf16")
x, noise = synthetic_images(64, seed=3)
x, noise = x.to(dev), noise.to(dev)
lib = capi.lib()
NAMES = {0: "all parts", 1: "no ActNorm sums", 14: "ActNorm sums only", 13: "conv2 only", 11: "conv3 only", 7: "conv1 only",
         12: "conv2 + ActNorm", 9: "conv2 + conv3", 5: "conv2 + conv1"}
for mask, label in NAMES.items():
    lib.nfk_debug_triple_parts(mask)
    for it in range(4):
        if it == 1:
            capi.profile_begin()
        for p in student.parameters():
            p.grad = None
        _, bpd, _ = student(x.clone(), None, noise=noise)
        bpd.mean().backward()
    prof = capi.profile_end()
    row = {d["kernel"][-22:]: round(d["ms"] * 1e3 / max(d["launches"], 1), 1) for d in prof if "triple" in d["kernel"] and d["launches"]}
    print(f"{label:22s}", row, flush=True)
lib.nfk_debug_triple_parts(0)
