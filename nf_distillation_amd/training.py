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
import torch

from . import capi
from .capi import check, ptr
from .dist import GradAllReducer
from .modules import FlowStep, SqueezeLayer


def kd_indices(student, teacher, student_is_1d=False):
    """NFModel._get_kd_indices (pl_module.py:81-110)."""
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

    def step(self, grad_scale=1.0):
        """grad_scale: e.g. 1/world_size after a sum all-reduce.  Returns nothing; the pre-clip global
        gradient norm is `self.grad_norm()` (device scalar, no sync)."""
        lib, s = capi.lib(), capi.stream()
        self.t += 1
        self.grad_scale = grad_scale
        self.sumsq.zero_()
        n = self.flat.numel
        check(lib.nfk_sumsq(ptr(self.flat.grad), n, ptr(self.partial), ptr(self.sumsq), s), "nfk_sumsq")
        check(lib.nfk_adam_step(ptr(self.flat.data), ptr(self.flat.grad), ptr(self.m), ptr(self.v), n,
                                ptr(self.sumsq), float(self.clip or 0.0), float(grad_scale), self.lr, self.betas[0],
                                self.betas[1], self.eps, self.wd, self.t, s), "nfk_adam_step")
        eng = getattr(self.flat.model.flow, "_engine", None)
        if eng is not None:
            eng.mark_dirty()

    def grad_norm(self):
        return torch.sqrt(self.sumsq[0]) * self.grad_scale


class DistillationStep:
    """One optimisation step of teacher->student distillation (see module docstring)."""

    def __init__(self, student, teacher, nll_weight=0.9, kd_weight=0.1, lr=5e-4, weight_decay=0.0,
                 clip_norm=30.0, process_group=None, bucket_bytes=16 << 20, student_is_1d=False):
        self.student, self.teacher = student, teacher
        self.w_nll, self.w_kd = nll_weight, kd_weight
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
        if self.world > 1:
            student.flow.engine().grad_hook = lambda idx: self.reducer.ready(self.flat.lo_for_layer(idx))

    def forward_losses(self, x, noise_student=None, noise_teacher=None):
        s_z, s_nll, _ = self.student(x, None, noise=noise_student)
        kd = None
        if self.w_kd > 0 and self.teacher is not None:
            with torch.no_grad():
                t_z, _, _ = self.teacher(x, None, noise=noise_teacher)
            for si, ti in zip(self.s_idx, self.t_idx):
                d = (s_z[si] - t_z[ti]).square().flatten(1).mean(1)
                kd = d if kd is None else kd + d
            kd = kd / max(len(self.s_idx), 1)
        else:
            kd = torch.zeros((), device=x.device)
        result = (self.w_nll * s_nll + self.w_kd * kd).mean()
        return {"nll": s_nll.mean(), "kd": kd.mean(), "result_loss": result}

    def training_step(self, x, noise_student=None, noise_teacher=None):
        """x: [B_local, C, H, W] on the GPU (mutated in place by the dequantisation noise, as upstream)."""
        self.flat.zero_grad()
        self.reducer.reset()
        out = self.forward_losses(x, noise_student, noise_teacher)
        out["result_loss"].backward()
        self.reducer.finish()
        self.opt.step(grad_scale=1.0 / self.world)
        return {k: v.detach() for k, v in out.items()}
