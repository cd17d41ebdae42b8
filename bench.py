#!/usr/bin/env python3
"""Headline benchmark: Glow CIFAR-10 teacher(K=32) -> student(K=8) distillation training_step.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default: bf16 operands / fp32 accumulate (BASELINE.json config 2), the whole step replayed from one hipGraph.
A "step" = teacher forward (no grad) + student forward/backward + MSE hints on layer pairs
[0,10,20,28]/[0,34,68,100] + global-norm clip 30 + Adam, on a synthetic
CIFAR-shaped batch of 64 images per GPU already resident in HBM.  Rank 0 prints ONE JSON line.

Extra objects on the line (rank 0, N=1 only):
  roofline        : K more (eager) steps with every GEMM-engine / fused-step launch bracketed by HIP events on
                    its own stream (nfk_profile_*); for the kernel class with the largest total time: algorithmic
                    FLOP/s against the dense MFMA peak of the compute dtype (157.3 TFLOP/s f32, 2.5 PFLOP/s bf16)
                    or algorithmic bytes/s against HBM (8 TB/s), whichever bound gives the larger minimum time.
  step_roofline   : the WHOLE step: 14.04 GFLOP per image x images / measured step time against the
                    same MFMA peak.
  f32_parity_mode : the same step in the reference's precision (the mode that meets the 1e-5 parity bar), timed the
                    same way, with its own dominant-kernel roofline and step roofline (skip: --no-other-precision).
  bpd_check       : student bits/dim of ONE batch (same images, same dequantisation noise, initial parameters) in
                    bf16 mode against fp32 mode; the run FAILS (exit 3) if they differ by 0.01 or more.
  cpu_baseline    : the CPU oracle (oracle/glow_oracle.py, a checked restatement of the reference)
                    running the same step on the host cores for a bounded number of steps.
  oracle_check    : THIS geometry against the oracle (tests/oracle_check.py): the cpu_baseline leg's warm-up step (B=64,
                    parameters of seeds 11 / 12, batch seed 100) is also run through the HIP path on the same batch and
                    noise -- default schedule (staged pipeline, one hipGraph) in both precisions, eager step in the headline one --
                    and step-0 nll / kd / loss / gradient norm / 64 sampled entries of every parameter gradient are
                    compared; the run FAILS (exit 5) on a miss.
  reference_precision_value / _ms_per_step : the f32 leg's numbers again, at top level (the reference computes in fp32;
                    `value` is BASELINE.json's bf16 mode).
Several ranks (N > 1): `ranks_seen` (dist.get_world_size()) and `allreduce_ms` (events around the bucketed gradient
all-reduce, averaged over the timed steps, max over ranks); `--backend gloo` is a stand-in transport for rehearsals on
one GPU (tests/test_gpu_bench_ranks.py).
The run also fails (exit 4) if the step could not be captured into a hipGraph: `config.schedule` is derived from the
runner's state, never from the flags.
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
GFLOP_PER_IMAGE = 14.04  # teacher fwd (8.023) + 3 x student fwd (2.006) per distillation step
PMC_PROFILE = "r03_b_pmc_traffic_bf16.json"   # profiles/: FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_traffic.py)


def cifar_cfg(K):
    return dict(image_shape=[32, 32, 3], hidden_channels=512, K=K, L=3, actnorm_scale=1.0,
                flow_permutation="invconv", flow_coupling="affine", LU_decomposed=True, y_classes=10,
                learn_top=False, y_condition=False, is_1d=False)


def make_models(device, dtype):
    import nf_distillation_amd as nfa
    from param_util import randomize_state_dict, shapes_of
    models = []
    for K, seed in ((8, 11), (32, 12)):
        m = nfa.create_glow_model(cifar_cfg(K))
        m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), seed))
        models.append(m.to(device).set_compute_dtype(dtype))
    return models  # student, teacher


def cpu_baseline(batch, steps, want_digest=False):
    """The oracle's distill_step on the host: 1 warm-up + `steps` timed steps.  The warm-up step (initial parameters,
    batch seed 100) doubles as the oracle side of `oracle_check`: -> (cpu_baseline block, digest of that step or None)."""
    from oracle import glow_oracle as O
    import oracle_check as OC
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
    digest = None
    for i in range(steps + 1):
        x, noise = synthetic_images(batch, seed=100 + i)
        t0 = time.perf_counter()
        out = O.distill_step(s_sd, s_cfg, t_sd, t_cfg, x, noise, noise.clone(), state, keep_grads=want_digest and i == 0)
        times.append(time.perf_counter() - t0)
        if want_digest and i == 0:
            digest = OC.digest_oracle_step(out, [k for k in s_sd if not O.is_buffer(k)])
    dt = sum(times[1:]) / steps
    return {"value": batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (after 1 warm-up) of the same distillation step, B={batch}, fp32, "
                      f"torch CPU {torch.__version__}"}, digest


def oracle_check(digest, batch, headline_dtype, both):
    """HIP step vs the oracle's step on the same batch / noise / parameters (tests/oracle_check.py)."""
    import oracle_check as OC
    cases = [(headline_dtype, "staged_graph"), (headline_dtype, "eager")]
    if both:
        cases.append(("f32" if headline_dtype == "bf16" else "bf16", "staged_graph"))
    block = {"sample": f"step 0 of the cpu_baseline leg: B={batch}, parameters of seeds {OC.SEED_STUDENT} / {OC.SEED_TEACHER}, "
                       "batch seed 100, same dequantisation noise", "bars": OC.BARS,
             "oracle": {k: digest[k] for k in ("nll", "kd", "result_loss", "grad_norm")}, "cases": {}}
    ok = True
    for dt, sched in cases:
        scal, grads, _ = OC.hip_step(dt, sched, batch, level_wgrad_aux=(dt == headline_dtype and sched != "eager"))
        res = OC.compare(digest, scal, grads, dt)
        res.update({k: scal[k] for k in ("nll", "kd", "result_loss", "grad_norm")})
        block["cases"][f"{dt}/{sched}"] = res
        ok = ok and res["ok"]
        del grads
        torch.cuda.empty_cache()
    block["ok"] = ok
    return block


class Leg:
    """One precision's models, step object and (captured) runner."""

    def __init__(self, args, dtype, device, data, world):
        from nf_distillation_amd.training import DistillationStep
        self.args, self.dtype, self.data, self.world = args, dtype, data, world
        self.student, self.teacher = make_models(device, dtype)
        if getattr(args, "bwd_mode", None) is not None:
            self.student.flow.engine().bwd_mode = args.bwd_mode
        if args.fused_runs is not None:
            self.teacher.flow.engine().fused_runs = args.fused_runs in ("both", "teacher")
            self.student.flow.engine().fused_runs = args.fused_runs in ("both", "student")
        # the level backward's per-level launches on the auxiliary stream: +3 % of the one-GPU step, bit-identical (engine.py)
        self.student.flow.engine().level_wgrad_aux = world == 1 and dtype == args.dtype
        if args.level_backward is not None:   # experiment knob: 0 per-step calls, 1 per level / one launch per step, 2 + runs, 3 + aux
            e = self.student.flow.engine()
            e.level_backward, e.level_multi, e.level_wgrad_aux = args.level_backward >= 1, args.level_backward >= 2, args.level_backward >= 3
        if args.wgrad_chunk is not None:
            self.student.flow.engine().level_wgrad_chunk = args.wgrad_chunk
        self.step = DistillationStep(self.student, self.teacher, nll_weight=0.9, kd_weight=0.1, lr=5e-4, clip_norm=30.0)
        if not args.no_overlap:
            self.step.enable_concurrency()  # teacher fwd || student fwd, wgrad || dgrad on side streams
        self.runner, self.pipelined, self.kind, self.fallbacks = self.step, False, "eager", []

    def initial_bpd(self, x, noise):
        """Student bits/dim of one batch with the initial parameters (no training)."""
        with torch.no_grad():
            _, bpd, _ = self.student(x.clone(), None, noise=noise)
        return float(bpd.mean())

    def _build_runner(self, kind):
        a, data = self.args, self.data
        if kind in ("ahead", "pipelined", "staged"):
            # software pipeline across steps: the frozen teacher's forward for batch t+1 runs on a side stream while the
            # student trains on batch t; every call still does one full step's work (nf_distillation_amd/pipeline.py).
            # "ahead": the teacher two batches ahead, its forward launched under the gradient all-reduce
            from nf_distillation_amd.pipeline import AheadPipelinedStep, PipelinedStep, StagedPipelinedStep
            cls = {"ahead": AheadPipelinedStep, "staged": StagedPipelinedStep}.get(kind, PipelinedStep)
            runner = cls(self.step, data[0], strict=True)
            runner.prepare(data[0])   # prime + eager warm-up + capture, before the W warm-up steps
        elif kind == "graph":
            from nf_distillation_amd.training import CapturedStep
            runner = CapturedStep(self.step, data[0], strict=True)
            runner.prepare()  # eager warm-up + capture, before the W warm-up steps
        else:
            self.step.set_allreduce_overlap(True)   # eager: buckets are all-reduced from inside the backward
            runner = self.step
        return runner

    def capture(self):
        a = self.args
        if not a.no_overlap:
            self.step.enable_concurrency()   # (the auxiliary stream is a library-wide setting: the other leg may have reset it)
        ahead = a.teacher_ahead == 2 or (a.teacher_ahead is None and self.world > 1)
        if a.no_graph:
            order = ["eager"]
        elif a.no_pipeline:
            order = ["graph", "eager"]
        else:
            staged = a.teacher_stages == 2 or (a.teacher_stages is None and self.world == 1)
            order = (["ahead"] if ahead else (["staged"] if staged else [])) + ["pipelined", "graph", "eager"]
        # one rank: a step that was not captured as asked is a failed run.  Several ranks (RCCL beside hipGraph capture has
        # never run in this repo's tests): degrade to the next simpler schedule and SAY so in config.schedule rather than
        # produce no number
        self.fallbacks = []
        for kind in order:
            try:
                if kind == "ahead":
                    # three graphs on two streams: with the auxiliary-stream branches inside the student's graph as well the
                    # graphs run one after the other (12.5 k images/s on one GPU instead of 18.2 k, profiles/r03_schedule_experiments.json)
                    self.student.flow.engine().level_wgrad_aux = False
                self.runner = self._build_runner(kind)
                self.kind = kind
                break
            except Exception as exc:   # pragma: no cover - depends on the runtime
                if self.world == 1:
                    print(f"bench.py: the step was not captured into a hipGraph ({kind}: {exc!r})", file=sys.stderr)
                    raise SystemExit(4)
                self.fallbacks.append(f"{kind}: {exc!r}"[:200])
                torch.cuda.synchronize()
        self.pipelined = self.kind in ("ahead", "pipelined", "staged")
        if self.world == 1 and not a.no_graph and getattr(self.runner, "graph", None) is None:
            print("bench.py: the step was not captured into a hipGraph", file=sys.stderr)
            raise SystemExit(4)

    def schedule(self):
        """What is actually running, read from the runner (never from the flags)."""
        graph = getattr(self.runner, "graph", None) is not None
        if graph and self.kind == "staged":
            s = ("hipGraph replay, software pipeline with the frozen teacher in two stages: teacher level 1 (t+2) || teacher deep "
                 "levels (t+1) || student (t)")
        elif graph and self.kind == "ahead":
            s = "hipGraph replay (three graphs), teacher two batches ahead: teacher(t+2) launched under all-reduce / optimiser (t)"
        elif graph and self.pipelined:
            s = "hipGraph replay, teacher(t+1) || student(t) software pipeline"
        elif graph:
            s = "hipGraph replay"
        else:
            s = "eager"
        if self.world > 1 and graph:
            s += ", gradient all-reduce between sub-graphs (never captured)"
        if self.fallbacks:
            s += " [fell back from: " + "; ".join(self.fallbacks) + "]"
        return s

    def run(self, n, eager=False):
        data, nb, last = self.data, len(self.data), None
        for i in range(n):
            if eager:
                last = self.step.training_step(data[i % nb].clone())
            elif self.pipelined:
                last = self.runner.training_step(data[(i + 1) % nb])   # trains on the batch supplied by the previous call
            else:
                last = self.runner.training_step(data[i % nb].clone())
        return last

    def timed(self, steps, warmup, device):
        from nf_distillation_amd.dist import max_over_ranks
        self.run(warmup)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = self.run(steps)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0, device), last

    def sequential_schedule(self, steps, warmup):
        """The same step without the cross-step software pipeline (teacher and student of the same batch inside one
        call, one hipGraph) -- reported beside the headline value, never instead of it."""
        from nf_distillation_amd.training import CapturedStep
        data, nb = self.data, len(self.data)
        seq = CapturedStep(self.step, data[0], strict=True)
        seq.prepare()
        for i in range(warmup):
            seq.training_step(data[i % nb].clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            seq.training_step(data[i % nb].clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": self.args.batch * steps / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / steps,
                "schedule": "hipGraph replay, teacher and student of the same batch per call"}

    def roofline(self, steps):
        """HIP events bracket each launch: eager steps (outside the captured graph), on ONE stream, so that a launch's
        duration is the kernel's own -- the same thing rocprofv3 --kernel-trace reports (profiles/r02_*)."""
        from nf_distillation_amd import capi
        self.step.enable_concurrency(False)
        if self.pipelined:   # both models run their deep-level fused steps with the pipelined schedule's split policy
            self.teacher.flow.engine().fused_split_inference = self.runner.teacher_split
            self.student.flow.engine().fused_split_training = self.runner.student_split
        self.run(2, eager=True)
        capi.profile_begin()
        self.run(steps, eager=True)
        prof = capi.profile_end()
        dom = max(prof, key=lambda d: d["ms"])
        n = max(dom["launches"], 1)
        avg_s = dom["ms"] * 1e-3 / n
        flops, nbytes = dom["flops"] / n, dom["bytes"] / n          # algorithmic, per launch
        mfma_peak = F32_MFMA_PEAK_TFLOPS if self.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
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
        pmc_file = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if self.dtype == "bf16" and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            name = pmc.get("bench_class_symbols", {}).get(dom["kernel"])
            kern = pmc.get("kernels", {})
            if name in kern:
                traffic, traffic_src = kern[name]["hbm_bytes_per_launch_corrected"], "profiles/" + PMC_PROFILE
        return {
            "bound": "hbm" if hbm_bound else "mfma",
            "kernel": dom["kernel"] + (" [bf16 operands]" if self.dtype == "bf16" else " [f32]"),
            "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "traffic": traffic,
            "traffic_source": traffic_src,
            "launches_per_step": dom["launches"] / steps, "avg_launch_us": 1e6 * avg_s,
            "algorithmic_gflop_per_launch": flops / 1e9, "algorithmic_mbytes_per_launch": nbytes / 1e6,
            "achieved_tflops": flops / avg_s / 1e12,
            "gemm_ms_per_step": {d["kernel"]: d["ms"] / steps for d in prof if d["launches"]},
        }


def step_roofline(images_per_s, dtype):
    peak = F32_MFMA_PEAK_TFLOPS if dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    tf = images_per_s * GFLOP_PER_IMAGE / 1e3
    return {"algorithmic_gflop_per_image": GFLOP_PER_IMAGE, "achieved_tflops": tf, "peak_tflops": peak, "frac": tf / peak,
            "note": "whole distillation step (all kernels, launch gaps, optimiser) against the dense MFMA peak"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling instead: fixed global batch (SURVEY.md §8d suggests 512), split evenly over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="skip the HIP-vs-oracle comparison of this geometry")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="transport of the gradient all-reduce for N > 1: nccl (= RCCL over xGMI, the product) or gloo "
                         "(a stand-in for rehearsing the N-rank path with several ranks on one GPU)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-precision", action="store_true",
                    help="skip the second leg (the f32 parity mode beside a bf16 headline, or the reverse)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true", help="keep every kernel on one stream")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run teacher and student of the SAME batch inside one call (the reference's order of work) "
                         "instead of the default software pipeline: teacher forward for batch t+1 beside the "
                         "student's step on batch t (nf_distillation_amd/pipeline.py; same values, every call still "
                         "does one teacher forward + one student forward/backward + one optimiser step)")
    ap.add_argument("--fused-runs", choices=("both", "teacher", "student", "off"), default=None,
                    help="experiment knob: which model walks a whole level's FlowSteps in ONE launch (levels 2 / 3; "
                         "default: the engines' default)")
    ap.add_argument("--level-backward", type=int, default=None, choices=(0, 1, 2, 3),
                    help="experiment knob: the student's bf16 backward -- 0 per-step calls, 1 per level with one launch per step, "
                         "2 with runs of steps in one launch, 3 with the per-level launches on the auxiliary stream (default: the engine's)")
    ap.add_argument("--wgrad-chunk", type=int, default=None, help="experiment knob: engine.level_wgrad_chunk of the student")
    ap.add_argument("--tune", type=int, nargs="*", default=None, metavar="KNOB VALUE",
                    help="experiment knobs: pairs for nfk_tune(knob, value) (include/nfk.h NFK_TUNE_*)")
    ap.add_argument("--bwd-mode", type=int, default=None, help="experiment knob: nfk_step2d_grads.bwd_mode of the student "
                                                                 "(1 separate launches, 3 one launch per conv, 4 / default fused)")
    ap.add_argument("--teacher-ahead", type=int, default=None, choices=(1, 2),
                    help="2: the teacher runs two batches ahead, its forward launched under the gradient all-reduce "
                         "(pipeline.AheadPipelinedStep).  Default: 2 for N > 1 (something overlaps the collective), 1 for "
                         "one GPU (no collective to hide; same speed)")
    ap.add_argument("--teacher-stages", type=int, default=None, choices=(1, 2),
                    help="2: the frozen teacher's forward as two pipeline stages beside each other (pipeline.StagedPipelinedStep: "
                         "level 1 of batch t+2 || deeper levels of batch t+1 || student of batch t); default: 2 on one GPU")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch from Python instead of replaying a hipGraph")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="bf16",
                    help="f32: the reference's precision (parity mode); bf16: bf16 operands / fp32 accumulate in the "
                         "coupling-net convs (BASELINE.json config 2; bits/dim within 0.01: tests/test_gpu_fullsize.py)")
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
    dev_index = local_rank
    if args.backend == "gloo":   # rehearsal: the ranks may share a GPU
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    from nf_distillation_amd import capi
    from param_util import synthetic_images

    tune = args.tune or []
    for knob, value in zip(tune[0::2], tune[1::2]):
        capi.lib().nfk_tune(knob, value)

    # synthetic data, resident in HBM before the timed region; a few distinct batches, cycled
    nb = 4
    data = [synthetic_images(args.batch, seed=1000 + 17 * rank + i)[0].to(device) for i in range(nb)]
    solo = rank == 0 and world == 1

    leg = Leg(args, args.dtype, device, data, world)
    other = None
    bpd_check = None
    if solo and not args.no_other_precision:
        other = Leg(args, "f32" if args.dtype == "bf16" else "bf16", device, data, world)
        xb, noise = synthetic_images(args.batch, seed=77)
        xb, noise = xb.to(device), noise.to(device)
        bpd = {leg.dtype: leg.initial_bpd(xb, noise), other.dtype: other.initial_bpd(xb, noise)}
        diff = abs(bpd["bf16"] - bpd["f32"])
        bpd_check = {"student_bpd_bf16": bpd["bf16"], "student_bpd_f32": bpd["f32"], "abs_diff": diff, "bar": 0.01,
                     "ok": diff < 0.01, "sample": f"one batch of {args.batch} images, same dequantisation noise, initial parameters"}

    leg.capture()
    if world > 1:
        leg.step.reducer.enable_timing()   # (after capture / warm-up of the runner: only timed + warm-up steps are averaged)
    elapsed, last = leg.timed(args.steps, args.warmup, device)
    images = args.batch * world * args.steps
    value = images / elapsed
    result = {
        "metric": "train images/sec Glow CIFAR-10 distillation step",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "Glow CIFAR-10 32x32 distillation training_step: teacher L=3 K=32 h=512 fwd (no grad) + "
                               "student L=3 K=8 h=512 fwd+bwd + MSE hints + clip 30 + Adam",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "schedule": leg.schedule()},
        "student_bpd": float(last["nll"]), "loss": float(last["result_loss"]),
        "step_roofline": step_roofline(value / world, args.dtype),
    }
    if world > 1:
        from nf_distillation_amd.dist import max_over_ranks
        result["ranks_seen"] = dist.get_world_size()
        result["backend"] = args.backend + (" (RCCL)" if args.backend == "nccl" else " (stand-in transport, not RCCL)")
        ar = leg.step.reducer.timed_ms()
        result["allreduce_ms"] = max_over_ranks(ar if ar is not None else 0.0, device)
        result["allreduce_note"] = ("events on the compute stream around GradAllReducer.finish(): what is left of the bucketed "
                                    "all-reduce after the backward has been enqueued + the wait for every bucket; max over ranks")
    if solo and leg.pipelined and not args.no_roofline:
        result["sequential_schedule"] = leg.sequential_schedule(args.steps, args.warmup)
    if solo and not args.no_roofline:
        result["roofline"] = leg.roofline(args.steps)
    if other is not None:
        other.capture()
        dt, last_o = other.timed(args.steps, args.warmup, device)
        v = args.batch * args.steps / dt
        block = {"value": v, "unit": "images/s", "ms_per_step": 1e3 * dt / args.steps, "dtype": other.dtype,
                 "schedule": other.schedule(), "student_bpd": float(last_o["nll"]), "step_roofline": step_roofline(v, other.dtype)}
        if not args.no_roofline:
            block["roofline"] = other.roofline(args.steps)
        result["f32_parity_mode" if other.dtype == "f32" else "bf16_throughput_mode"] = block
        result["bpd_check"] = bpd_check
    f32_block = result if args.dtype == "f32" else result.get("f32_parity_mode")
    if f32_block is not None:
        result["reference_precision_value"] = f32_block["value"]
        result["reference_precision_ms_per_step"] = f32_block["ms_per_step"]
    result["precision_note"] = (
        "the reference computes in fp32: `reference_precision_value` (images/s, exact-f32 MFMA, meets the 1e-5 parity bar) is "
        "the number to hold against it; `value` is BASELINE.json's bf16 mode (bf16 operands, fp32 accumulation; student "
        "bits/dim within 0.01)" if args.dtype == "bf16" else "fp32 throughout (the reference's precision)")
    check_block = None
    if solo and not args.no_cpu_baseline:
        want = not args.no_oracle_check
        # (the legs and their captured graphs stay alive to the end of the process: see engine.level_wgrad_aux)
        result["cpu_baseline"], digest = cpu_baseline(args.batch, args.cpu_steps, want_digest=want)
        if want:
            check_block = result["oracle_check"] = oracle_check(digest, args.batch, args.dtype, not args.no_other_precision)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if bpd_check is not None and not bpd_check["ok"]:
        print("bench.py: bf16 student bits/dim differs from fp32 by >= 0.01", file=sys.stderr)
        raise SystemExit(3)
    if check_block is not None and not check_block["ok"]:
        print("bench.py: the HIP step at the benchmark's geometry disagrees with the oracle (oracle_check)", file=sys.stderr)
        raise SystemExit(5)


if __name__ == "__main__":
    main()
