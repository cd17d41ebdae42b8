#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one MI355X (synthetic data of the configured shape, parameters
randomised as in the parity tests), with the roofline each is priced against.  usage: python tools/config_bench.py out.json

  * CelebA-64 Glow L=4 K=32 h=512 (config 4): one training step (forward + backward + clip + Adam), bf16 operands
  * CelebA-64 distillation, teacher L=3 K=32 h=512 -> student L=3 K=16 h=256 (conf/{teacher,student}/celeba.yaml)
  * 1-D Glow d=6 K=5 h=32 (POWER, config 1) and d=63 K=10 h=32 (BSDS300, config 5): training step and forward at
    B = 1024 and 65 536 rows; HBM roofline with 2 * B * d * 4 algorithmic bytes per FlowStep
"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import nf_distillation_amd as nfa  # noqa: E402
from nf_distillation_amd import capi  # noqa: E402
from nf_distillation_amd.training import CapturedStep, DistillationStep  # noqa: E402
from param_util import randomize_state_dict, shapes_of  # noqa: E402

BF16_PEAK, HBM_PEAK = 2500.0, 8000.0


def glow_cfg(shape, K, L, hid, is_1d=False):
    return dict(image_shape=shape, hidden_channels=hid, K=K, L=L, actnorm_scale=1.0, flow_permutation="invconv",
                flow_coupling="affine", LU_decomposed=True, y_classes=10, learn_top=False, y_condition=False, is_1d=is_1d)


def build(cfg, seed, dev, dtype):
    m = nfa.create_glow_model(cfg)
    if cfg.get("flow_coupling") == "made":   # (MADE blocks: masks are buffers of the model; parameters as tests/test_gpu_made.py)
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, v in m.state_dict().items():
            if ".mask" in k:
                sd[k] = v.clone()
            elif k.endswith("weight"):
                sd[k] = torch.randn(v.shape, generator=g) * (0.6 / v.shape[1] ** 0.5)
            else:
                sd[k] = torch.randn(v.shape, generator=g) * 0.1
        m.load_state_dict(sd)
    else:
        m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), seed))
    return m.to(dev).set_compute_dtype(dtype)


def fwd_gflop_per_image(cfg):
    h, w, c = cfg["image_shape"]
    hid, K, L = cfg["hidden_channels"], cfg["K"], cfg["L"]
    tot, C, H, W = 0.0, c, h, w
    for lvl in range(L):
        C, H, W = C * 4, H // 2, W // 2
        hw = H * W
        tot += K * (2 * hw * (9 * (C // 2) * hid + hid * hid + 9 * hid * C) + 2 * hw * C * C)
        if lvl < L - 1:
            tot += 2 * 9 * (C // 2) * C * hw
            C //= 2
    return tot / 1e9


def time_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def image_case(name, s_cfg, t_cfg, batch, dev, steps=20):
    student = build(s_cfg, 21, dev, "bf16")
    teacher = build(t_cfg, 22, dev, "bf16") if t_cfg else None
    step = DistillationStep(student, teacher, nll_weight=0.9 if teacher else 1.0, kd_weight=0.1 if teacher else 0.0, lr=5e-4,
                            clip_norm=30.0)
    if teacher is not None:
        step.enable_concurrency()
    h, w, c = s_cfg["image_shape"]
    x = (torch.randint(0, 256, (batch, c, h, w), device=dev).float() / 256 - 0.5)
    if teacher is not None:   # bench.py's default schedule on one GPU: the frozen teacher as two pipeline stages
        from nf_distillation_amd.pipeline import StagedPipelinedStep
        runner = StagedPipelinedStep(step, x, strict=True)
        runner.prepare(x)
        dt = time_steps(lambda: runner.training_step(x), steps, 3)
        sched = "hipGraph replay, staged software pipeline (teacher level 1 (t+2) || teacher deep levels (t+1) || student (t))"
    else:
        runner = CapturedStep(step, x, strict=True)
        runner.prepare()
        dt = time_steps(lambda: runner.training_step(x.clone()), steps, 3)
        sched = "hipGraph replay"
    gflop = 3 * fwd_gflop_per_image(s_cfg) + (fwd_gflop_per_image(t_cfg) if t_cfg else 0.0)
    tf = batch * gflop / dt / 1e3
    fused_fwd = [int(capi.lib().nfk_step2d_fused_tiles(__import__("ctypes").byref(r.cs)) > 0)
                 for r in student.flow.engine().state["recs"] if r.kind == "step"]
    return {"config": name, "batch": batch, "images_per_s": batch / dt, "ms_per_step": 1e3 * dt, "schedule": sched,
            "algorithmic_gflop_per_image": gflop, "achieved_tflops": tf, "roofline": {"bound": "mfma", "peak_tflops": BF16_PEAK,
                                                                                    "frac": tf / BF16_PEAK},
            "student_steps_on_fused_forward": f"{sum(fused_fwd)} of {len(fused_fwd)}"}


def tab_case(d, K, hid, batch, dev, made=False):
    cfg = glow_cfg([d], K, 1, hid, is_1d=True)
    if made:
        cfg["flow_coupling"] = "made"
    model = build(cfg, 31, dev, "f32")
    step = DistillationStep(model, None, nll_weight=1.0, kd_weight=0.0, lr=5e-4, clip_norm=30.0)
    x = torch.randn(batch, d, device=dev)
    runner = CapturedStep(step, x, strict=True)
    runner.prepare()
    steps = 200 if batch <= 4096 else 50
    dt_train = time_steps(lambda: runner.training_step(x), steps, 5)

    def fwd():
        with torch.no_grad():
            model(x, None)
    dt_fwd = time_steps(fwd, steps, 5)
    bytes_fwd = K * 2 * batch * d * 4
    flop_row = K * (2 * ((d // 2) * hid + 4 * hid * hid + hid * 2 * (d - d // 2)) + 2 * d * d)
    return {"config": (f"MAF / MADE blocks d={d} K={K} h={hid} (parity unpinned: no reference code)" if made
                       else f"1-D Glow d={d} K={K} h={hid}"), "batch_rows": batch, "train_rows_per_s": batch / dt_train,
            "train_ms_per_step": 1e3 * dt_train, "train_schedule": "hipGraph replay (forward + backward + clip + Adam)",
            "forward_rows_per_s": batch / dt_fwd, "forward_ms": 1e3 * dt_fwd, "forward_schedule": "eager, K launches",
            "roofline_forward": {"bound": "hbm", "algorithmic_bytes": bytes_fwd, "achieved_GBps": bytes_fwd / dt_fwd / 1e9,
                                 "peak_GBps": HBM_PEAK, "frac": bytes_fwd / dt_fwd / 1e9 / HBM_PEAK},
            "forward_gflops": batch * flop_row / dt_fwd / 1e9}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/configs.json"
    dev = torch.device("cuda", 0)
    rows = []
    rows.append(image_case("CIFAR-10 Glow L=3 K=8 h=512, student-only training step, bf16 (BASELINE config 2)",
                           glow_cfg([32, 32, 3], 8, 3, 512), None, 64, dev, steps=30))
    print(rows[-1], flush=True)
    torch.cuda.empty_cache()
    rows.append(image_case("CelebA-64 Glow L=4 K=32 h=512, training step (BASELINE config 4)", glow_cfg([64, 64, 3], 32, 4, 512),
                           None, 16, dev, steps=10))
    print(rows[-1], flush=True)
    torch.cuda.empty_cache()
    rows.append(image_case("CelebA-64 distillation: teacher L=3 K=32 h=512 -> student L=3 K=16 h=256", glow_cfg([64, 64, 3], 16, 3, 256),
                           glow_cfg([64, 64, 3], 32, 3, 512), 16, dev, steps=10))
    print(rows[-1], flush=True)
    torch.cuda.empty_cache()
    for d, K in ((6, 5), (63, 10)):
        for batch in (1024, 65536):
            rows.append(tab_case(d, K, 32, batch, dev))
            print(rows[-1], flush=True)
            rows.append(tab_case(d, K, 32, batch, dev, made=True))
            print(rows[-1], flush=True)
    json.dump({"what": "1 x MI355X; see tools/config_bench.py", "rows": rows}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
