"""Host-side mirror of the reference's distillation ``training_step``.

Restates, for the hot path only, what ``pl_module.NFModel`` + ``pl.Trainer`` do per step:
  forward            pl_module.py:198-255  student(x) with grad, teacher(x) under no_grad on the
                                           SAME in-place-noised x
  loss               pl_module.py:257-320  per-sample MSE hints on paired layers / n_pairs,
                                           w_nll * nll + w_kd * kd, mean over the batch
  _get_kd_indices    pl_module.py:81-110
  optimiser          pl_module.py:350-354  Adam(student.parameters(), lr, weight_decay)
  clipping           train.py:46           gradient_clip_val=30 (global L2 norm)
with the student's parameters and gradients re-homed into flat fp32 arenas so that clipping,
Adam and the RCCL gradient all-reduce each touch ONE contiguous buffer.
"""
import ctypes as C

import torch

from . import capi
from .capi import check, ptr
from .dist import GradAllReducer
from .modules import FlowStep, SqueezeLayer


def kd_indices(student, teacher, student_is_1d=None):
    """NFModel._get_kd_indices (pl_module.py:81-110).  `student_is_1d` is params["student"]["is_1d"] upstream; by
    default it is read from the student model."""
    if student_is_1d is None:
        student_is_1d = bool(getattr(student, "is_1d", False))

    def pick(model, mult):
        layers = model.flow.layers
        return [i for i, l in enumerate(layers)
                if isinstance(l, SqueezeLayer) or (student_is_1d and (i + 1) % mult == 0) or i + 1 == len(layers)]
    return pick(student, 2), pick(teacher, 4)


class FlatParams:
    """All parameters of `model` as views into one flat fp32 arena (and `.grad` into another).

    Layout: [late region | layer-ordered region].  "Late" parameters get their gradient only at
    the very end of the backward (ActNorm / invconv factors, chained by one nfk_prep_bwd launch;
    learn_top / y_condition heads); the rest is in layer order, so during backward the arena
    completes from the END towards the front -- what GradAllReducer.ready() expects.
    Every segment starts on a 16-byte boundary (float4 kernels)."""

    def __init__(self, model):
        named = list(model.named_parameters())
        late, ordered = [], []
        for name, p in named:
            in_layer = name.startswith("flow.layers.")
            is_net = ".block." in name or (in_layer and name.split(".")[3] == "conv")
            (ordered if (in_layer and is_net) else late).append((name, p))
        self.entries = []
        off = 0
        self.layer_lo = {}
        for name, p in late + ordered:
            n = p.numel()
            self.entries.append((name, p, off, n))
            if name.startswith("flow.layers.") and (name, p) in ordered:
                li = int(name.split(".")[2])
                self.layer_lo.setdefault(li, off)
            off += (n + 3) & ~3
        self.numel = off
        dev = named[0][1].device
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for name, p, o, n in self.entries:
            if p.dtype != torch.float32:
                raise TypeError(f"{name}: fp32 parameters expected")
            view = self.data[o:o + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + n].view(p.shape)
        self.model = model

    def attach_grads(self):
        """Re-attach the flat grad views (an external zero_grad(set_to_none=True) detaches them)."""
        for _, p, o, n in self.entries:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        self.attach_grads()

    def lo_for_layer(self, idx):
        """Smallest arena offset owned by layers >= idx (everything at/after it is final once
        layer idx has run its backward)."""
        cands = [o for li, o in self.layer_lo.items() if li >= idx]
        return min(cands) if cands else self.numel


class FlatAdam:
    """clip_grad_norm_ + Adam over a FlatParams arena in three launches (nfk_sumsq x2, nfk_adam_step)."""

    def __init__(self, flat, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_norm=30.0):
        self.flat, self.lr, self.betas, self.eps, self.wd, self.clip = flat, lr, betas, eps, weight_decay, clip_norm
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.sumsq = torch.zeros(4, device=flat.data.device, dtype=torch.float32)
        self.partial = torch.zeros(1024, device=flat.data.device, dtype=torch.float32)
        self.t = 0
        # completed-step counter on the device: every step issues the identical launch sequence (graph-replayable)
        self.step_dev = torch.zeros(4, device=flat.data.device, dtype=torch.int32)

    def step(self, grad_scale=1.0):
        """grad_scale: e.g. 1/world_size after a sum all-reduce.  Returns nothing; the pre-clip global
        gradient norm is `self.grad_norm()` (device scalar, no sync)."""
        lib, s = capi.lib(), capi.stream()
        self.t += 1
        self.grad_scale = grad_scale
        self.sumsq.zero_()
        n = self.flat.numel
        check(lib.nfk_sumsq(ptr(self.flat.grad), n, ptr(self.partial), ptr(self.sumsq), s), "nfk_sumsq")
        check(lib.nfk_adam_step_dev(ptr(self.flat.data), ptr(self.flat.grad), ptr(self.m), ptr(self.v), n,
                                    ptr(self.sumsq), float(self.clip or 0.0), float(grad_scale), self.lr,
                                    self.betas[0], self.betas[1], self.eps, self.wd, ptr(self.step_dev), s),
              "nfk_adam_step_dev")
        eng = getattr(self.flat.model.flow, "_engine", None)
        if eng is not None:
            eng.mark_dirty()

    def grad_norm(self):
        return torch.sqrt(self.sumsq[0]) * self.grad_scale


def _dense_pair(s, t):
    """(s, t) hint tensors must be dense, same shape and same memory order (both are channels-last views of the
    engine's NHWC buffers, or both contiguous)."""
    if s.shape != t.shape or s.stride() != t.stride() or s.dtype != torch.float32 or t.dtype != torch.float32:
        raise ValueError("hint tensors must be fp32 with identical shapes and strides")
    B = s.shape[0]
    per = s.numel() // max(B, 1)
    dense = s.is_contiguous() or (s.dim() == 4 and s.is_contiguous(memory_format=torch.channels_last))
    if not dense or (B > 1 and s.stride(0) != per):
        raise ValueError("hint tensors must be dense and per-sample contiguous")
    return per


class _KdLossFn(torch.autograd.Function):
    """result = mean_b(w_nll * nll[b] + w_kd * kd[b]),  kd[b] = mean over hint pairs of the per-sample MSE
    (NFModel.loss, pl_module.py:246-300) -- two launches forward, one backward (nfk_kd_loss_fwd / _bwd) instead of
    the ~45 elementwise / reduction kernels the same expression costs as tensor ops."""

    @staticmethod
    def forward(ctx, s_nll, w_nll, w_kd, n_pairs, divisor, weights, *tensors):
        s_list, t_list = tensors[:n_pairs], tensors[n_pairs:]
        capi.require_cuda(s_nll, "s_nll")
        B = s_nll.shape[0]
        if n_pairs > capi.KD_MAX_PAIRS:
            raise ValueError(f"at most {capi.KD_MAX_PAIRS} hint pairs are supported, got {n_pairs}")
        pairs = capi.KdPairs()
        pairs.n, pairs.B, pairs.divisor = n_pairs, B, max(int(divisor), n_pairs)
        if weights is not None:
            weights = weights.detach().to(torch.float32).reshape(B).contiguous()
            pairs.weights = weights.data_ptr()
        for i, (s, t) in enumerate(zip(s_list, t_list)):
            pairs.per_sample[i] = _dense_pair(s, t)
            pairs.s[i], pairs.t[i] = s.data_ptr(), t.data_ptr()
        nll = s_nll.detach().to(torch.float32).contiguous()
        part = torch.empty(max(n_pairs, 1) * B, device=nll.device, dtype=torch.float32)
        kd = torch.empty(B, device=nll.device, dtype=torch.float32)
        out3 = torch.empty(3, device=nll.device, dtype=torch.float32)
        capi.check(capi.lib().nfk_kd_loss_fwd(C.byref(pairs), capi.ptr(nll), float(w_nll), float(w_kd), capi.ptr(part),
                                              capi.ptr(kd), capi.ptr(out3), capi.stream()), "nfk_kd_loss_fwd")
        ctx.pairs, ctx.w = pairs, (float(w_nll), float(w_kd))
        ctx.keep = (s_list, t_list, nll, weights)
        ctx.n_pairs = n_pairs
        ctx.mark_non_differentiable(kd)
        ctx.set_materialize_grads(False)   # (no zero tensors for the three outputs nothing differentiates: 3 fill launches)
        return out3[0], out3[1].detach(), out3[2].detach(), kd

    @staticmethod
    def backward(ctx, g_result, _g1, _g2, _g3):
        s_list, _, nll, weights = ctx.keep
        if g_result is None:
            return (None,) * (6 + 2 * len(s_list))
        g = g_result.detach().to(torch.float32).reshape(1).contiguous()
        grads = capi.KdGrads()
        g_s = [torch.empty_like(s) for s in s_list]   # preserves the (channels-last) strides
        for i, t in enumerate(g_s):
            grads.g_s[i] = t.data_ptr()
        g_nll = torch.empty_like(nll)
        grads.g_nll = g_nll.data_ptr()
        if ctx.n_pairs > 0:
            capi.check(capi.lib().nfk_kd_loss_bwd(C.byref(ctx.pairs), capi.ptr(g), ctx.w[0], ctx.w[1], C.byref(grads),
                                                  capi.stream()), "nfk_kd_loss_bwd")
        else:
            g_nll = g * (ctx.w[0] / nll.shape[0]) * (weights if weights is not None else torch.ones_like(nll))
        return (g_nll, None, None, None, None, None) + tuple(g_s) + (None,) * len(g_s)


def distillation_loss(s_nll, s_hints, t_hints, w_nll, w_kd, weights=None):
    """-> dict(nll, kd, result_loss) as NFModel.loss logs them; only result_loss carries a gradient.
    The two hint lists are ZIPPED as upstream (pl_module.py:264-267: the shorter list decides the number of pairs)
    while the divisor stays len(student hints) (:279-281); student hints without a partner get no gradient.
    `weights`: optional per-sample weights multiplied into the loss before the batch mean (:309-310)."""
    n = min(len(s_hints), len(t_hints))
    result, nll_m, kd_m, _ = _KdLossFn.apply(s_nll, w_nll, w_kd, n, len(s_hints), weights, *s_hints[:n],
                                             *[t.detach() for t in t_hints[:n]])
    return {"nll": nll_m, "kd": kd_m, "result_loss": result}


class DistillationStep:
    """One optimisation step of teacher->student distillation (see module docstring)."""

    def __init__(self, student, teacher, nll_weight=0.9, kd_weight=0.1, lr=5e-4, weight_decay=0.0,
                 clip_norm=30.0, process_group=None, bucket_bytes=16 << 20, student_is_1d=None,
                 perceptual_weight=0.0):
        self.student, self.teacher = student, teacher
        self.w_nll, self.w_kd, self.w_perc = nll_weight, kd_weight, perceptual_weight
        if perceptual_weight > 0 and (teacher is None or not getattr(student, "is_1d", False)):
            # images use VGG features upstream (pl_module.py:178: VGGPerceptualLoss on pretrained weights -- outside the hot
            # path, and weight 0 in every image config).  The reverse pass itself carries gradients for image flows too
            # (engine._Decode2dFn): a caller can put any loss on model(z=..., reverse=True)[-1].
            raise ValueError("the perceptual (L1) branch is the tabular one: it needs 1-D flows and a teacher")
        if teacher is not None:
            for p in teacher.parameters():
                p.requires_grad_(False)
            teacher.eval()
            self.s_idx, self.t_idx = kd_indices(student, teacher, student_is_1d)
        else:
            self.s_idx, self.t_idx = [], []
        self.flat = FlatParams(student)
        self.opt = FlatAdam(self.flat, lr=lr, weight_decay=weight_decay, clip_norm=clip_norm)
        self.reducer = GradAllReducer(self.flat.grad, process_group, bucket_bytes)
        self.world = self.reducer.world
        self._side = self._aux = self._aux_handle = None
        self._concurrent = False
        self.set_allreduce_overlap(True)

    def enable_concurrency(self, on=True):
        """Run the (frozen, no-grad) teacher forward on a side stream concurrently with the student forward, and
        each FlowStep's weight-gradient GEMMs on an auxiliary stream concurrently with its dgrad chain.  The
        launches are small (64-512 workgroups on 256 CUs), so independent branches fill otherwise idle CUs.
        Results are bitwise unchanged: only the order in which independent kernels may run differs."""
        if on and self._side is None:
            self._side = torch.cuda.Stream()
            self._aux = torch.cuda.Stream()
        self._concurrent = bool(on)
        if on and self._aux_handle is None:
            self._aux_handle = capi.lib().nfk_aux_create(self._aux.cuda_stream)
        eng = self.student.flow.engine() if hasattr(self.student, "flow") else None
        if eng is not None:
            eng.aux = self._aux_handle if on else None
            eng.aux_stream = self._aux if on else None

    def _cond(self, y_onehot):
        """pl_module.py:216-243: y goes to the models iff params["student"]["y_condition"]."""
        if getattr(self.student, "y_condition", False):
            if y_onehot is None:
                raise ValueError("the student is conditional (y_condition): training_step needs y_onehot")
            return y_onehot
        return None

    def _teacher_forward(self, x, noise_teacher, y_onehot=None):
        # KD-aware encode: of the teacher's 29 / 101 layer outputs the loss reads the hint layers only
        # (pl_module.py:262-282), so only those are materialised (the rest of the list is None)
        with torch.no_grad():
            t_z, _, _ = self.teacher(x, y_onehot, noise=noise_teacher, keep=self.t_idx, hints_only=True)
        return t_z

    def perceptual_l1(self, x, latent=None, y_onehot=None):
        """pl_module.py:229-243, 283-299 for tabular data: student (with gradients) and teacher decode the same latent
        ~ student prior at temperature 0.7; per-sample mean |x_s - x_t|, NaN rows zeroed."""
        from .modules import gaussian_sample
        if latent is None:
            mean, logs = self.student.prior(x, y_onehot)
            latent = gaussian_sample(mean, logs, 1)
        s_x = self.student(z=latent, temperature=0.7, reverse=True, y_onehot=y_onehot)[-1]
        with torch.no_grad():
            t_x = self.teacher(z=latent, temperature=0.7, reverse=True, y_onehot=y_onehot)[-1]
        per = (s_x - t_x).abs().flatten(1).mean(1)
        return torch.where(torch.isnan(per), torch.zeros_like(per), per)

    def forward_losses(self, x, noise_student=None, noise_teacher=None, latent=None, y_onehot=None, weights=None):
        y = self._cond(y_onehot)
        out = self._nll_kd_losses(x, noise_student, noise_teacher, y, weights)
        if self.w_perc > 0:
            per = self.perceptual_l1(x, latent, y)
            out["perceptual"] = per.mean()
            # mean_b of a sum = sum of the means; the per-sample weights multiply every term (pl_module.py:301-310)
            out["result_loss"] = out["result_loss"] + self.w_perc * (per if weights is None else per * weights).mean()
        return out

    def _nll_kd_losses(self, x, noise_student=None, noise_teacher=None, y=None, weights=None):
        use_teacher = self.w_kd > 0 and self.teacher is not None
        t_z = None
        if use_teacher and self._concurrent:
            # The teacher reads (and re-dequantises, in place) the SAME x right after the student's first kernel
            # has consumed it (pl_module.py:216-225 order): fork there.
            main = torch.cuda.current_stream()
            box = {}
            eng = self.student.flow.engine()

            def fork():
                ev = torch.cuda.Event()
                ev.record(main)
                self._side.wait_event(ev)
                with torch.cuda.stream(self._side):
                    box["t_z"] = self._teacher_forward(x, noise_teacher, y)
                done = torch.cuda.Event()
                done.record(self._side)
                box["done"] = done

            eng.first_layer_hook = fork
            try:
                s_z, s_nll, _ = self.student(x, y, noise=noise_student)
            finally:
                eng.first_layer_hook = None
            main.wait_event(box["done"])
            t_z = box["t_z"]
        else:
            s_z, s_nll, _ = self.student(x, y, noise=noise_student)
            if use_teacher:
                t_z = self._teacher_forward(x, noise_teacher, y)
        if use_teacher:
            return distillation_loss(s_nll, [s_z[i] for i in self.s_idx], [t_z[i] for i in self.t_idx],
                                     self.w_nll, self.w_kd, weights)
        return distillation_loss(s_nll, [], [], self.w_nll, 0.0, weights)

    # A step = three phases; multi-rank graph replay captures the first and the last separately so that the
    # collective itself is never part of a captured graph.
    def fwd_bwd(self, x, noise_student=None, noise_teacher=None, latent=None, y_onehot=None, weights=None):
        self.flat.zero_grad()
        self.reducer.reset()
        out = self.forward_losses(x, noise_student, noise_teacher, latent, y_onehot, weights)
        out["result_loss"].backward()
        return {k: v.detach() for k, v in out.items()}

    def allreduce(self):
        self.reducer.finish()   # launches whatever the backward hooks have not launched yet, then waits

    def optimize(self):
        self.opt.step(grad_scale=1.0 / self.world)

    def set_allreduce_overlap(self, on):
        """on: buckets are all-reduced from inside the backward (eager mode); off: once, after it (graph mode)."""
        eng = self.student.flow.engine()
        eng.grad_hook = (lambda idx: self.reducer.ready(self.flat.lo_for_layer(idx))) if (on and self.world > 1) else None

    def training_step(self, x, noise_student=None, noise_teacher=None, latent=None, y_onehot=None, weights=None):
        """x: [B_local, C, H, W] on the GPU (mutated in place by the dequantisation noise, as upstream), or [B, d] rows.
        latent: optional caller-drawn latent of the perceptual branch (tests); default = a draw from the student prior.
        y_onehot: the batch's condition (required iff the student has y_condition; pl_module.py:216-243).
        weights: optional per-sample loss weights (the `drop_weights` data sets, pl_module.py:309-310)."""
        out = self.fwd_bwd(x, noise_student, noise_teacher, latent, y_onehot, weights)
        self.allreduce()
        self.optimize()
        return out


class CapturedStep:
    """The training_step replayed from hipGraphs instead of being enqueued launch by launch from Python.

    A step issues several hundred small launches; enqueueing them from the host costs as much time as the GPU
    needs to run them.  After `warmup` eager steps (which also fix every lazily built descriptor and kernel
    attribute) the step is captured with torch.cuda.graph and then replayed; the host only copies the batch
    into a static buffer and calls hipGraphLaunch.
      * one rank : ONE graph = zero-grad + student fwd + teacher fwd + losses + backward + clip + Adam
      * N ranks  : graph A (zero-grad .. backward)  ->  eager bucketed RCCL all-reduce of the flat gradient
                   arena  ->  graph B (clip + Adam).  No collective is ever captured.
    Falls back to eager execution (with a warning) if capture fails, unless `strict`; `self.graph is not None` tells
    which of the two is running.
    """

    def __init__(self, step, example_x, warmup=2, with_noise=False, example_y=None, example_weights=None, strict=False):
        self.step = step
        self.strict = strict    # True: a failed capture raises instead of degrading to eager execution
        self.static_x = example_x.clone()
        # condition / per-sample weights of the batch (conditional students, weighted data sets): static buffers too
        self.static_y = None if example_y is None else example_y.clone()
        self.static_w = None if example_weights is None else example_weights.clone()
        if self.static_y is None and getattr(step.student, "y_condition", False):
            raise ValueError("the student is conditional (y_condition): CapturedStep needs example_y")
        # optional caller-supplied dequantisation noise (parity tests); otherwise drawn on the device inside the graph
        self.static_ns = torch.zeros_like(example_x) if with_noise else None
        self.static_nt = torch.zeros_like(example_x) if with_noise else None
        self.graph = self.graph_b = None
        self.out = None
        self.enabled = True
        self._warm = warmup
        self.split = step.world > 1
        if self.split:
            step.set_allreduce_overlap(False)

    def prepare(self):
        """Run the eager warm-up steps and capture now (so that no timed step pays for it)."""
        while self.enabled and self.graph is None:
            self.training_step(self.static_x.clone(), self.static_ns, self.static_nt, self.static_y, self.static_w)

    def _eager(self):
        return self.step.training_step(self.static_x, self.static_ns, self.static_nt, None, self.static_y, self.static_w)

    def _capture(self):
        torch.cuda.synchronize()
        if not self.split:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.out = self._eager()
            self.graph = g
            return
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            self.out = self.step.fwd_bwd(self.static_x, self.static_ns, self.static_nt, None, self.static_y, self.static_w)
        with torch.cuda.graph(gb, pool=ga.pool()):
            self.step.optimize()
        self.graph, self.graph_b = ga, gb

    def _replay(self):
        self.graph.replay()
        if self.split:
            self.step.reducer.reset()
            self.step.allreduce()
            self.graph_b.replay()
        return self.out

    def training_step(self, x, noise_student=None, noise_teacher=None, y_onehot=None, weights=None):
        if not self.enabled:
            return self.step.training_step(x, noise_student, noise_teacher, None, y_onehot, weights)
        self.static_x.copy_(x)
        if self.static_ns is not None and noise_student is not self.static_ns:
            self.static_ns.copy_(noise_student)
            self.static_nt.copy_(noise_teacher)
        if self.static_y is not None and y_onehot is not self.static_y:
            self.static_y.copy_(y_onehot)
        if self.static_w is not None and weights is not self.static_w:
            self.static_w.copy_(weights)
        if self.graph is None:
            if self._warm > 0:
                self._warm -= 1
                return self._eager()
            try:
                self._capture()
                return self._replay()   # capturing does not execute: run the step it describes
            except Exception as exc:  # pragma: no cover - depends on the runtime
                if self.strict:
                    raise
                import warnings
                warnings.warn(f"hipGraph capture failed ({exc!r}); running eagerly")
                self.enabled = False
                self.graph = self.graph_b = None
                if self.split:
                    self.step.set_allreduce_overlap(True)
                torch.cuda.synchronize()
                return self.step.training_step(x, noise_student, noise_teacher, None, y_onehot, weights)
        return self._replay()
