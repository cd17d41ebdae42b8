#!/usr/bin/env python3
"""1-GPU batch sweep of the headline step (Glow CIFAR-10 teacher K=32 -> student K=8 distillation, bf16 operands), both
schedules, written as JSON: what strong scaling at a fixed global batch would leave every rank with (8 ranks at the
reference's global batch of 64 -> 8 images per GPU; at 512 -> 64).  usage: python tools/batch_sweep.py out.json"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
from param_util import synthetic_images  # noqa: E402


def run(batch, pipelined, steps, warmup, dev):
    ns = argparse.Namespace(no_overlap=False, no_graph=False, no_pipeline=not pipelined, batch=batch, fused_runs=None, dtype="bf16",
                            level_backward=None, wgrad_chunk=None, teacher_ahead=None, teacher_stages=None, bwd_mode=None)
    data = [synthetic_images(batch, seed=1000 + i)[0].to(dev) for i in range(4)]
    leg = bench.Leg(ns, "bf16", dev, data, 1)
    leg.capture()
    dt, _ = leg.timed(steps, warmup, dev)
    return {"per_gpu_batch": batch, "schedule": leg.schedule(), "images_per_s": batch * steps / dt, "ms_per_step": 1e3 * dt / steps}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/batch_sweep.json"
    dev = torch.device("cuda", 0)
    rows = []
    for batch in (8, 16, 32, 64, 128, 256):
        for pipelined in (True, False):
            steps = 40 if batch <= 64 else 20
            r = run(batch, pipelined, steps, 5, dev)
            rows.append(r)
            print(r, flush=True)
            torch.cuda.empty_cache()
    by = {(r["per_gpu_batch"], "pipeline" in r["schedule"]): r for r in rows}
    pred = {}
    for gb in (64, 512):
        pred[f"global_batch_{gb}"] = {
            f"{n}_gpus": {"per_gpu_batch": gb // n,
                          "compute_only_images_per_s": n * by[(gb // n, True)]["images_per_s"] if (gb // n, True) in by else None}
            for n in (1, 2, 4, 8)}
    doc = {"what": "1 x MI355X, bf16 operands, hipGraph replay; compute-only prediction of strong scaling = n x (1-GPU rate at "
                   "the per-GPU batch); the gradient all-reduce (44 MB fp32) is NOT in it: unmeasured on hardware",
           "rows": rows, "strong_scaling_compute_only": pred}
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
