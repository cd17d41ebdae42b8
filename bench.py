#!/usr/bin/env python3
"""Headline benchmark: Glow CIFAR-10 teacher(K=32) -> student(K=8) distillation training_step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default: bf16 operands / fp32 accumulate (BASELINE.json config 2), the whole step replayed from one hipGraph.
A "step" = teacher forward (no grad) + student forward/backward + MSE hints on layer pairs
[0,10,20,28]/[0,34,68,100] + global-norm clip 30 + Adam, on a synthetic
CIFAR-shaped batch of 64 images per GPU already resident in HBM.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N=1 only):
  roofline     : K more (eager) steps with every GEMM-engine / fused-step launch bracketed by HIP events on
                 its own stream (nfk_profile_*); for the kernel class with the largest total time: algorithmic
                 FLOP/s against the dense MFMA peak of the compute dtype (157.3 TFLOP/s f32, 2.5 PFLOP/s bf16)
                 or algorithmic bytes/s against HBM (8 TB/s), whichever bound gives the larger minimum time.
  cpu_baseline : the CPU oracle (oracle/glow_oracle.py, a checked restatement of the reference)
                 running the same step on the host cores for a bounded number of steps.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: Peak BF16 MFMA, dense
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak (spec)


def cifar_cfg(K):
    return dict(image_shape=[32, 32, 3], hidden_channels=512, K=K, L=3, actnorm_scale=1.0,
                flow_permutation="invconv", flow_coupling="affine", LU_decomposed=True, y_classes=10,
                learn_top=False, y_condition=False, is_1d=False)


def make_models(device):
    import nf_distillation_amd as nfa
    from param_util import randomize_state_dict, shapes_of
    models = []
    for K, seed in ((8, 11), (32, 12)):
        m = nfa.create_glow_model(cifar_cfg(K))
        m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), seed))
        models.append(m.to(device))
    return models  # student, teacher


def cpu_baseline(batch, steps):
    """The oracle's distill_step on the host: 1 warm-up + `steps` timed steps."""
    from oracle import glow_oracle as O
    from param_util import randomize_state_dict, synthetic_images
    from test_oracle_golden import full_shapes
    # a 1-GPU box gets a 16-CPU share of the host; more threads than that only thrash
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    s_cfg, t_cfg = cifar_cfg(8), cifar_cfg(32)
    s_sd = randomize_state_dict(full_shapes(s_cfg), 11)
    t_sd = randomize_state_dict(full_shapes(t_cfg), 12)
    state = {}
    times = []
    for i in range(steps + 1):
        x, noise = synthetic_images(batch, seed=100 + i)
        t0 = time.perf_counter()
        O.distill_step(s_sd, s_cfg, t_sd, t_cfg, x, noise, noise.clone(), state)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / steps
    return {"value": batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (after 1 warm-up) of the same distillation step, B={batch}, fp32, "
                      f"torch CPU {torch.__version__}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling instead: fixed global batch (SURVEY.md §8d suggests 512), split evenly over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true", help="keep every kernel on one stream")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run teacher and student of the SAME batch inside one call (the reference's order of work) "
                         "instead of the default software pipeline: teacher forward for batch t+1 beside the "
                         "student's step on batch t (nf_distillation_amd/pipeline.py; same values, every call still "
                         "does one teacher forward + one student forward/backward + one optimiser step)")
    ap.add_argument("--tuning", type=int, default=None, help="experiment knob: nfk_set_tuning(v)")
    ap.add_argument("--bwd-mode", type=int, default=None, help="experiment knob: nfk_set_bwd_mode(v) (0: dgrad and wgrad as "
                                                                 "separate launches, 2: one launch per conv, default)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch from Python instead of replaying a hipGraph")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="bf16",
                    help="f32: the reference's precision (parity mode); bf16: bf16 operands / fp32 accumulate in the "
                         "coupling-net convs (BASELINE.json config 2; bits/dim within 0.01: tests/test_gpu_bf16.py)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    strong = args.global_batch is not None
    if strong:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of ranks")
        args.batch = args.global_batch // world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI

    from nf_distillation_amd.dist import max_over_ranks
    from nf_distillation_amd.training import DistillationStep
    from nf_distillation_amd import capi
    from param_util import synthetic_images

    if args.tuning is not None:
        capi.lib().nfk_set_tuning(args.tuning)
    if args.bwd_mode is not None:
        capi.lib().nfk_set_bwd_mode(args.bwd_mode)
    student, teacher = make_models(device)
    student.set_compute_dtype(args.dtype)
    teacher.set_compute_dtype(args.dtype)
    step = DistillationStep(student, teacher, nll_weight=0.9, kd_weight=0.1, lr=5e-4, clip_norm=30.0)
    if not args.no_overlap:
        step.enable_concurrency()  # teacher fwd || student fwd, wgrad || dgrad on side streams

    # synthetic data, resident in HBM before the timed region; a few distinct batches, cycled
    nb = 4
    data = [synthetic_images(args.batch, seed=1000 + 17 * rank + i)[0].to(device) for i in range(nb)]

    # one hipGraph for the whole step (fwd + bwd + clip + Adam); the host only copies the batch in and replays
    runner = step
    pipelined = not args.no_graph and not args.no_pipeline
    if pipelined:
        # software pipeline across steps: the frozen teacher's forward for batch t+1 runs on a side stream while the
        # student trains on batch t; every call still does one full step's work (nf_distillation_amd/pipeline.py)
        from nf_distillation_amd.pipeline import PipelinedStep
        runner = PipelinedStep(step, data[0])
        runner.prepare(data[0])   # prime + eager warm-up + capture, before the W warm-up steps
    elif not args.no_graph:
        from nf_distillation_amd.training import CapturedStep
        runner = CapturedStep(step, data[0])
        runner.prepare()  # eager warm-up + capture, before the W warm-up steps

    def run(n, eager=False):
        last = None
        for i in range(n):
            if eager:
                last = step.training_step(data[i % nb].clone())
            elif pipelined:
                last = runner.training_step(data[(i + 1) % nb])   # trains on the batch supplied by the previous call
            else:
                last = runner.training_step(data[i % nb].clone())
        return last

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0, device)

    images = args.batch * world * args.steps
    result = {
        "metric": "train images/sec Glow CIFAR-10 distillation step",
        "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "Glow CIFAR-10 32x32 distillation training_step: teacher L=3 K=32 h=512 fwd (no grad) + "
                               "student L=3 K=8 h=512 fwd+bwd + MSE hints + clip 30 + Adam",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "schedule": ("hipGraph replay, teacher(t+1) || student(t) software pipeline" if pipelined else
                                ("eager" if args.no_graph else "hipGraph replay"))},
        "student_bpd": float(last["nll"]), "loss": float(last["result_loss"]),
    }

    if rank == 0 and world == 1 and pipelined and not args.no_roofline:
        # for reference: the same step without the cross-step software pipeline (teacher and student of the same batch
        # inside one call, one hipGraph) -- reported beside the headline value, never instead of it
        from nf_distillation_amd.training import CapturedStep
        seq = CapturedStep(step, data[0])
        seq.prepare()
        for i in range(args.warmup):
            seq.training_step(data[i % nb].clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            seq.training_step(data[i % nb].clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        result["sequential_schedule"] = {"value": args.batch * args.steps / dt, "unit": "images/s",
                                         "ms_per_step": 1e3 * dt / args.steps,
                                         "schedule": "hipGraph replay, teacher and student of the same batch per call"}

    if rank == 0 and world == 1 and not args.no_roofline:
        # HIP events bracket each launch: eager steps (outside the captured graph), on ONE stream, so that a launch's
        # duration is the kernel's own -- the same thing rocprofv3 --kernel-trace reports (profiles/r01_h_*)
        step.enable_concurrency(False)
        if pipelined:   # both models run their deep-level fused steps with the pipelined schedule's split policy
            teacher.flow.engine().fused_split_inference = runner.teacher_split
            student.flow.engine().fused_split_training = runner.student_split
        run(2, eager=True)
        capi.profile_begin()
        run(args.steps, eager=True)
        prof = capi.profile_end()
        dom = max(prof, key=lambda d: d["ms"])
        n = max(dom["launches"], 1)
        avg_s = dom["ms"] * 1e-3 / n
        flops, nbytes = dom["flops"] / n, dom["bytes"] / n          # algorithmic, per launch
        mfma_peak = F32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
        # the roofline that bounds this kernel is the one with the larger minimum time
        hbm_bound = nbytes / (HBM_PEAK_GBPS * 1e9) > flops / (mfma_peak * 1e12)
        if hbm_bound:
            achieved, peak, unit = nbytes / avg_s / 1e9, HBM_PEAK_GBPS, "GB/s"
        else:
            achieved, peak, unit = flops / avg_s / 1e12, mfma_peak, "TFLOP/s"
        # HBM bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE with the gfx950
        # fetch correction, tools/pmc_traffic.py) -- bench.py cannot collect counters itself; null if the profile does
        # not hold this kernel
        traffic, traffic_src = None, None
        sym = {"<*,1,*>": "nfk::fused2_step_fwd_kernel<3, 1, 4, 1>", "<*,2,*>": "nfk::fused2_step_fwd_kernel<2, 2, 8, 4>",
               "<*,3,*>": "nfk::fused2_step_fwd_kernel<2, 3, 16, 4>"}   # levels 2 / 3: the student's instances
        pmc_file = os.path.join(ROOT, "profiles", "r01_h_pmc_traffic_bf16.json")
        if args.dtype == "bf16" and os.path.exists(pmc_file):
            kern = json.load(open(pmc_file)).get("kernels", {})
            for tag, name in sym.items():
                if tag in dom["kernel"] and name in kern:
                    traffic, traffic_src = kern[name]["hbm_bytes_per_launch_corrected"], "profiles/r01_h_pmc_traffic_bf16.json"
        result["roofline"] = {
            "bound": "hbm" if hbm_bound else "mfma",
            "kernel": dom["kernel"] + (" [bf16 operands]" if args.dtype == "bf16" else " [f32]"),
            "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
            "launches_per_step": dom["launches"] / args.steps, "avg_launch_us": 1e6 * avg_s,
            "algorithmic_gflop_per_launch": flops / 1e9, "algorithmic_mbytes_per_launch": nbytes / 1e6,
            "achieved_tflops": flops / avg_s / 1e12,
            "gemm_ms_per_step": {d["kernel"]: d["ms"] / args.steps for d in prof},
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.batch, args.cpu_steps)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
