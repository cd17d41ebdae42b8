"""The benchmark's own geometry against the oracle: ONE distillation step of the CIFAR pair (teacher L=3 K=32 h=512 ->
student L=3 K=8 h=512, B=64 per GPU, the parameters bench.py times: seeds 11 / 12) on the CPU oracle
(oracle/glow_oracle.py `distill_step`, restating /root/reference/pl_module.py:198-320,350-354 and train.py:46) and on the
HIP path, on the SAME batch and the SAME dequantisation noise, through the schedules bench.py runs.

Shared by tests/test_gpu_bench_geometry.py and by bench.py's `oracle_check` block (which reuses the warm-up step of its
`cpu_baseline` leg as the oracle side).  Test infrastructure: nothing in the package imports this.

Bars
  f32  : nll, kd, result_loss within 1e-5 relative; pre-clip global gradient norm within 1e-3 relative; 64 sampled entries
         of EVERY parameter gradient within 1e-3 of that tensor's largest (reference) entry.
  bf16 : |nll - oracle nll| < 0.01 bits/dim (BASELINE.json's bar); cosine between all sampled gradient entries and the
         oracle's > 0.999; gradient norm within 2 %.
"""
import torch

from golden_util import N_SAMPLES, sample_index
from param_util import randomize_state_dict, synthetic_images

SEED_STUDENT, SEED_TEACHER = 11, 12   # bench.py make_models / cpu_baseline
BARS = {"f32": {"loss_rel": 1e-5, "grad_norm_rel": 1e-3, "sample_rel_to_max": 1e-3},
        "bf16": {"nll_abs": 0.01, "grad_norm_rel": 2e-2, "sampled_cosine": 0.999}}


def cifar_cfg(K):
    return dict(image_shape=[32, 32, 3], hidden_channels=512, K=K, L=3, actnorm_scale=1.0,
                flow_permutation="invconv", flow_coupling="affine", LU_decomposed=True, y_classes=10,
                learn_top=False, y_condition=False, is_1d=False)


def check_batch(batch, seed=100):
    """(x, noise_student, noise_teacher) on the CPU: what bench.py's cpu_baseline feeds its first step."""
    x, noise = synthetic_images(batch, seed=seed)
    return x, noise, noise.clone()


def digest_oracle_step(out, names):
    """Scalars + per-tensor (largest |entry|, 64 sampled entries) of the oracle's UNCLIPPED gradients.
    `out` = oracle.distill_step(..., keep_grads=True): its gradients are those after clip_grad_norm_."""
    gn = float(out["grad_norm"])
    coef = min(1.0, 30.0 / (gn + 1e-6))
    maxes, samples = [], []
    for i, k in enumerate(names):
        v = out["grads"][k].detach().double().flatten() / coef
        idx = sample_index(v.numel(), i)
        s = torch.zeros(N_SAMPLES, dtype=torch.float64)
        s[: idx.numel()] = v[idx]
        maxes.append(float(v.abs().max()))
        samples.append(s)
    return {"nll": float(out["nll"]), "kd": float(out["kd"]), "result_loss": float(out["result_loss"]), "grad_norm": gn,
            "names": list(names), "max": maxes, "samples": torch.stack(samples)}


def oracle_step(batch, seed=100, threads=None):
    """One oracle step at the bench geometry on the host cores -> digest (see digest_oracle_step)."""
    from oracle import glow_oracle as O
    from test_oracle_golden import full_shapes
    if threads:
        torch.set_num_threads(threads)
    s_cfg, t_cfg = cifar_cfg(8), cifar_cfg(32)
    s_sd = randomize_state_dict(full_shapes(s_cfg), SEED_STUDENT)
    t_sd = randomize_state_dict(full_shapes(t_cfg), SEED_TEACHER)
    x, ns, nt = check_batch(batch, seed)
    out = O.distill_step(s_sd, s_cfg, t_sd, t_cfg, x, ns, nt, {}, keep_grads=True)
    return digest_oracle_step(out, [k for k in s_sd if not O.is_buffer(k)])


class OptState:
    """Snapshot / restore of everything a training step changes (parameters, Adam moments, step counter), in place:
    lets a captured graph be warmed up and captured on the real buffers and then replayed from the initial state."""

    def __init__(self, step):
        self.step = step
        o = step.opt
        self.bufs = [step.flat.data, o.m, o.v, o.step_dev]
        self.saved = [b.clone() for b in self.bufs]
        self.t = o.t

    def restore(self):
        for b, s in zip(self.bufs, self.saved):
            b.copy_(s)
        self.step.opt.t = self.t
        self.step.student.flow.engine().mark_dirty()
        torch.cuda.synchronize()


_KEEP = []   # runners (and their captured graphs) stay alive for the life of the process (engine.level_wgrad_aux)


def hip_step(dtype, schedule, batch, seed=100, device="cuda", level_wgrad_aux=False):
    """The same step through libnfk.  schedule: "eager" (DistillationStep.training_step, one stream), "graph" (the whole
    step as one hipGraph) or "pipelined_graph" (bench.py's default: teacher(t+1) || student(t), one hipGraph).
    -> (scalars dict, {name: unclipped gradient on the device}, schedule actually run)"""
    import nf_distillation_amd as nfa
    from nf_distillation_amd.pipeline import PipelinedStep
    from nf_distillation_amd.training import CapturedStep, DistillationStep
    from param_util import shapes_of
    models = []
    for K, sd_seed in ((8, SEED_STUDENT), (32, SEED_TEACHER)):
        m = nfa.create_glow_model(cifar_cfg(K))
        m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), sd_seed))
        models.append(m.to(device).set_compute_dtype(dtype))
    step = DistillationStep(models[0], models[1], nll_weight=0.9, kd_weight=0.1, lr=5e-4, clip_norm=30.0)
    models[0].flow.engine().level_wgrad_aux = bool(level_wgrad_aux)   # (bench.py's headline leg runs with it)
    x, ns, nt = (t.to(device) for t in check_batch(batch, seed))
    ran = schedule
    if schedule == "eager":
        out = step.training_step(x.clone(), ns, nt)
    else:
        step.enable_concurrency()
        state = OptState(step)
        if schedule == "graph":
            runner = CapturedStep(step, x, warmup=2, with_noise=True, strict=True)
            for _ in range(3):   # 2 eager warm-ups + capture (+ first replay)
                runner.training_step(x.clone(), ns, nt)
            assert runner.graph is not None, "capture did not happen"
            state.restore()
            out = runner.training_step(x.clone(), ns, nt)
        elif schedule == "staged_graph":   # bench.py's default on one GPU: the teacher as two pipeline stages
            from nf_distillation_amd.pipeline import StagedPipelinedStep
            runner = StagedPipelinedStep(step, x, use_graph=True, warmup=2, with_noise=True, strict=True)
            runner.prepare(x, ns, nt)
            assert runner.graph is not None, "capture did not happen"
            state.restore()
            runner.prime(x, x, ns, nt)
            out = runner.training_step(x, ns, nt)   # trains on the first primed batch
        elif schedule == "pipelined_graph":
            runner = PipelinedStep(step, x, use_graph=True, warmup=2, with_noise=True, strict=True)
            runner.prepare(x, ns, nt)
            assert runner.graph is not None, "capture did not happen"
            state.restore()
            runner.prime(x, ns, nt)
            out = runner.training_step(x, ns, nt)   # trains on the primed batch
        else:
            raise ValueError(schedule)
        out = {k: v.clone() for k, v in out.items()}
        _KEEP.append(runner)
    torch.cuda.synchronize()
    scal = {k: float(out[k]) for k in ("nll", "kd", "result_loss")}
    scal["grad_norm"] = float(step.opt.grad_norm())
    grads = {name: step.flat.grad[o:o + n].clone() for name, prm, o, n in step.flat.entries}
    step.enable_concurrency(False)
    return scal, grads, ran


def compare(ref, scal, grads, dtype):
    """-> dict of error figures + "ok" against BARS[dtype]."""
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    worst, worst_name = 0.0, None
    num = da = db = 0.0
    for i, k in enumerate(ref["names"]):
        v = grads[k].detach().double().flatten().cpu()
        idx = sample_index(v.numel(), i)
        got, r = v[idx], ref["samples"][i][: idx.numel()]
        num, da, db = num + float(got @ r), da + float(got @ got), db + float(r @ r)
        if ref["max"][i] > 0:
            e = float((got - r).abs().max()) / ref["max"][i]
            if e > worst:
                worst, worst_name = e, k
    res = {
        "nll_rel": rel(scal["nll"], ref["nll"]), "nll_abs": abs(scal["nll"] - ref["nll"]),
        "kd_rel": rel(scal["kd"], ref["kd"]), "result_loss_rel": rel(scal["result_loss"], ref["result_loss"]),
        "grad_norm_rel": rel(scal["grad_norm"], ref["grad_norm"]),
        "worst_sample_rel_to_tensor_max": worst, "worst_tensor": worst_name,
        "sampled_cosine": num / max(da ** 0.5 * db ** 0.5, 1e-300), "tensors": len(ref["names"]),
    }
    bar = BARS[dtype]
    if dtype == "f32":
        ok = (max(res["nll_rel"], res["kd_rel"], res["result_loss_rel"]) <= bar["loss_rel"] and
              res["grad_norm_rel"] <= bar["grad_norm_rel"] and worst <= bar["sample_rel_to_max"])
    else:
        ok = (res["nll_abs"] < bar["nll_abs"] and res["grad_norm_rel"] <= bar["grad_norm_rel"] and
              res["sampled_cosine"] > bar["sampled_cosine"])
    res["ok"] = bool(ok)
    return res
