"""Diagnostic: 50-step training trajectories of the full-size distillation pair in fp32, fp32 with a 1e-7 relative
perturbation of the initial parameters (how fast does fp32 rounding-level noise grow under training?), and bf16."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import Golden  # noqa: E402
from param_util import synthetic_images  # noqa: E402
from test_gpu_fullsize import _distill_pair  # noqa: E402

g = Golden("distill_full")
B, steps = 16, 50
out = {}
for tag, dt, eps in (("f32", "f32", 0.0), ("f32_perturbed", "f32", 1e-7), ("bf16", "bf16", 0.0)):
    step = _distill_pair(g, dt)
    if eps:
        gen = torch.Generator(device="cuda").manual_seed(5)
        step.flat.data.mul_(1.0 + eps * torch.randn(step.flat.data.shape, device="cuda", generator=gen))
        step.student.flow.engine().mark_dirty()
    vals, evals = [], []
    for i in range(steps):
        x, _ = synthetic_images(B, (3, 32, 32), seed=500 + i % 5)
        gen = torch.Generator().manual_seed(900 + i)
        ns = (torch.rand(x.shape, generator=gen) / 256.0).cuda()
        nt = (torch.rand(x.shape, generator=gen) / 256.0).cuda()
        vals.append(float(step.training_step(x.cuda(), ns, nt)["nll"]))
    out[tag] = vals
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "trajectory_probe.json"), "w"))
for i in range(steps):
    print(i, " ".join(f"{out[k][i]:.5f}" for k in out), f"{out['f32_perturbed'][i]-out['f32'][i]:+.5f} {out['bf16'][i]-out['f32'][i]:+.5f}")
