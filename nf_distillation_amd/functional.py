"""Stand-alone forwards of sub-modules that the hot path normally runs fused inside FlowStep."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import capi
from .capi import PrepDesc, check, ptr


def conv2d_zeros_small(mod, h):
    """Conv2dZeros on the prior tensor [1, 2C, H, W] (learn_top, models/flows.py:382-383).
    Batch-independent, parameter-only work on a single "image": done with device tensor ops."""
    w = mod.conv.weight
    y = F.conv2d(h, w, mod.conv.bias, 1, (w.shape[2] // 2, w.shape[3] // 2))
    return y * torch.exp(mod.logs * mod.logscale_factor)


def _mix(x, C_, pixels, is_1d, logdet, reverse, **desc_fields):
    """prep + chanmix for one lone ActNorm / InvertibleConv1x1 (forward only, no autograd)."""
    if torch.is_grad_enabled() and (x.requires_grad or any(
            torch.is_tensor(v) and v.requires_grad for v in desc_fields.values())):
        raise NotImplementedError("stand-alone ActNorm/InvertibleConv1x1 calls are inference-only; "
                                  "gradients flow through FlowStep / FlowNet")
    capi.require_cuda(x)
    lib, s = capi.lib(), capi.stream()
    dev = x.device
    buf = torch.zeros(2 * C_ * C_ + 2 * C_ + 8 + 5 * C_ * C_, device=dev)
    zeros = torch.zeros(C_, device=dev)
    d = PrepDesc()
    d.C, d.pixels, d.is_1d = C_, pixels, 1 if is_1d else 0
    d.an_bias = d.an_logs = ptr(zeros)
    keep = [zeros]
    for k, v in desc_fields.items():
        if torch.is_tensor(v):
            v = v.detach().contiguous()
            keep.append(v)
            setattr(d, k, ptr(v))
        else:
            setattr(d, k, v)
    base = buf.data_ptr()
    o = 0
    for nm, n in (("Wf", C_ * C_), ("cf", C_), ("Wi", C_ * C_), ("ci", C_), ("dld", 4), ("scratch", 5 * C_ * C_)):
        setattr(d, nm, base + 4 * o)
        o += (n + 3) & ~3
    dd = capi.struct_array_to_device([d], dev)
    check(lib.nfk_prep(ptr(dd), 1, 1 if reverse else 0, s), "nfk_prep")
    xx = x.contiguous() if is_1d else x.permute(0, 2, 3, 1).contiguous()
    out = torch.empty_like(xx)
    Mrows = xx.numel() // C_
    Wm, cv = (d.Wi, d.ci) if reverse else (d.Wf, d.cf)
    check(lib.nfk_chanmix(ptr(xx), C_, Wm, cv, ptr(out), C_, Mrows, C_, s), "nfk_chanmix")
    dld = buf[(d.dld - base) // 4]
    if logdet is not None:
        logdet = logdet - dld if reverse else logdet + dld
    return (out if is_1d else out.permute(0, 3, 1, 2)), logdet


def actnorm_apply(mod, x, logdet, reverse):
    C_ = mod.num_features
    idx = torch.arange(C_, device=x.device, dtype=torch.int32)
    pixels = 1 if mod.is_1d else x.shape[2] * x.shape[3]
    return _mix(x, C_, pixels, mod.is_1d, logdet, reverse, perm_kind=2, perm_idx=idx,
                an_bias=mod.bias.reshape(-1), an_logs=mod.logs.reshape(-1))


def invconv_apply(mod, x, logdet, reverse):
    C_ = mod.w_shape[0]
    pixels = 1 if mod.is_1d else x.shape[2] * x.shape[3]
    if mod.LU_decomposed:
        return _mix(x, C_, pixels, mod.is_1d, logdet, reverse, perm_kind=0, lower=mod.lower, upper=mod.upper,
                    log_s=mod.log_s, p=mod.p, sign_s=mod.sign_s)
    return _mix(x, C_, pixels, mod.is_1d, logdet, reverse, perm_kind=1, weight=mod.weight)
