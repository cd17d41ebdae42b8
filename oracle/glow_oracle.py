"""CPU oracle for the Glow coupling-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the algorithm the
reference implements in ``models/layers.py``, ``models/flows.py``, ``models/kd_flows.py``
and ``models/utils.py`` plus the loss/optimiser policy of ``pl_module.py`` / ``train.py``.
It exists so that the HIP path can be checked against something that is itself pinned to
the reference: ``tests/test_oracle_golden.py`` compares every function here with golden
vectors captured from the reference's own code (``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.  The product package ``nf_distillation_amd`` never does, and it
has no CPU fallback: it raises when the HIP library is missing.

The oracle is *functional*: a model is a ``dict`` config (the reference's
``create_glow_model`` kwargs, ``models/kd_flows.py:76-105``) plus a flat ``state_dict``
whose keys/shapes equal the reference's (SURVEY.md §5 "Checkpoint / resume").
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# layer schedule (models/flows.py:234-269)
# --------------------------------------------------------------------------------------
def layer_schedule(cfg) -> List[Tuple[str, tuple]]:
    """[(kind, output_shape_without_batch)], kind in {"squeeze","step","split"}."""
    is_1d = bool(cfg.get("is_1d", False))
    sched = []
    if is_1d:
        c = cfg["image_shape"][0]
        for _ in range(cfg["L"] * cfg["K"]):
            sched.append(("step", (c,)))
        return sched
    h, w, c = cfg["image_shape"]
    for lvl in range(cfg["L"]):
        c, h, w = c * 4, h // 2, w // 2
        sched.append(("squeeze", (c, h, w)))
        for _ in range(cfg["K"]):
            sched.append(("step", (c, h, w)))
        if lvl < cfg["L"] - 1:
            sched.append(("split", (c // 2, h, w)))
            c //= 2
    return sched


def cond_features(cfg) -> int:
    return cfg["y_classes"] if cfg.get("y_condition", False) else 0


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def gaussian_logp(mean, logs, x):
    """models/layers.py:10-23; `logs` is log-std.  Returns per-sample sum."""
    lp = -0.5 * (2.0 * logs + (x - mean) ** 2 * torch.exp(-2.0 * logs) + LN_2PI)
    return lp.flatten(1).sum(1)


def squeeze2(x):
    """models/layers.py:32-44 (factor 2): out[b, 4c+2i+j, h, w] = in[b, c, 2h+i, 2w+j]."""
    b, c, h, w = x.shape
    assert h % 2 == 0 and w % 2 == 0
    return (
        x.reshape(b, c, h // 2, 2, w // 2, 2)
        .permute(0, 1, 3, 5, 2, 4)
        .reshape(b, 4 * c, h // 2, w // 2)
    )


def unsqueeze2(x):
    """models/layers.py:47-61 (factor 2)."""
    b, c, h, w = x.shape
    assert c % 4 == 0
    return (
        x.reshape(b, c // 4, 2, 2, h, w)
        .permute(0, 1, 4, 2, 5, 3)
        .reshape(b, c // 4, 2 * h, 2 * w)
    )


def _pixels(x):
    return 1 if x.dim() == 2 else x.shape[2] * x.shape[3]


def actnorm(bias, logs, x, logdet, reverse):
    """models/layers.py:101-142.  bias/logs are [1,C,1,1] or [1,C]."""
    dld = logs.sum() * _pixels(x)
    if not reverse:
        y = (x + bias) * torch.exp(logs)
        ld = None if logdet is None else logdet + dld
    else:
        y = x * torch.exp(-logs) - bias
        ld = None if logdet is None else logdet - dld
    return y, ld


def invconv_matrix(sd, pfx, reverse):
    """models/layers.py:360-402.  Returns (W, sum(log|s|))."""
    if pfx + "weight" in sd:
        w = sd[pfx + "weight"]
        dld = torch.slogdet(w)[1]
        return (torch.inverse(w) if reverse else w), dld
    lower, upper = sd[pfx + "lower"], sd[pfx + "upper"]
    p, sign_s, log_s = sd[pfx + "p"], sd[pfx + "sign_s"], sd[pfx + "log_s"]
    n = lower.shape[0]
    eye = torch.eye(n, dtype=lower.dtype)
    lo = torch.tril(lower, -1) + eye
    up = torch.triu(upper, 1) + torch.diag(sign_s * torch.exp(log_s))
    if reverse:
        w = torch.inverse(up) @ (torch.inverse(lo) @ torch.inverse(p))
    else:
        w = p @ (lo @ up)
    return w, log_s.sum()


def invconv(sd, pfx, x, logdet, reverse):
    """models/layers.py:404-421.  2-D: z = W x per pixel; 1-D: z = x @ W."""
    w, s = invconv_matrix(sd, pfx, reverse)
    dld = s * _pixels(x)
    if x.dim() == 2:
        z = x @ w
    else:
        z = torch.einsum("oc,bchw->bohw", w, x)
    if logdet is not None:
        logdet = logdet - dld if reverse else logdet + dld
    return z, logdet


def conv_actnorm(sd, pfx, x):
    """Conv2d with SAME zero padding, no bias, followed by ActNorm2d (models/layers.py:190-228)."""
    w = sd[pfx + "conv.weight"]
    y = F.conv2d(x, w, None, 1, (w.shape[2] // 2, w.shape[3] // 2))
    return (y + sd[pfx + "actnorm.bias"]) * torch.exp(sd[pfx + "actnorm.logs"])


def conv_zeros(sd, pfx, x):
    """Conv2dZeros: SAME conv with bias, times exp(3*logs) (models/layers.py:231-260)."""
    w = sd[pfx + "conv.weight"]
    y = F.conv2d(x, w, sd[pfx + "conv.bias"], 1, (w.shape[2] // 2, w.shape[3] // 2))
    return y * torch.exp(sd[pfx + "logs"] * 3.0)


def linear_zeros(sd, pfx, x):
    """LinearZeros (models/layers.py:173-187)."""
    return F.linear(x, sd[pfx + "linear.weight"], sd[pfx + "linear.bias"]) * torch.exp(
        sd[pfx + "logs"] * 3.0
    )


def coupling_net(sd, pfx, x, is_1d):
    """get_block_2d / get_block_1d (models/flows.py:25-52)."""
    if not is_1d:
        h = torch.relu(conv_actnorm(sd, pfx + "0.", x))
        h = torch.relu(conv_actnorm(sd, pfx + "2.", h))
        return conv_zeros(sd, pfx + "4.", h)
    h = x
    for i in (0, 2, 4, 6):
        h = torch.relu(F.linear(h, sd[f"{pfx}{i}.weight"], sd[f"{pfx}{i}.bias"]))
    h = torch.tanh(F.linear(h, sd[pfx + "8.weight"], sd[pfx + "8.bias"]))
    return F.linear(h, sd[pfx + "10.weight"], sd[pfx + "10.bias"])


def permute_channels(x, indices, reverse):
    """Permute2d (models/layers.py:263-290); `indices` is the forward gather index."""
    if reverse:
        inv = torch.empty_like(indices)
        inv[indices] = torch.arange(indices.numel())
        indices = inv
    return x[:, indices]


def flow_step(sd, cfg, pfx, x, logdet, y_onehot, reverse, perm_indices=None):
    """FlowStep.normal_flow / reverse_flow (models/flows.py:142-202)."""
    is_1d = bool(cfg.get("is_1d", False))
    perm = cfg.get("flow_permutation", "invconv")
    coupling = cfg.get("flow_coupling", "affine")
    red = [1] if is_1d else [1, 2, 3]

    def permutation(t, ld, rev):
        if perm == "invconv":
            return invconv(sd, pfx + "invconv.", t, ld, rev)
        return permute_channels(t, perm_indices, rev), ld

    def net(z1):
        # forward conditions on "y given", reverse on "model has condition features"
        # (models/flows.py:152 vs :177)
        use_y = (y_onehot is not None) if not reverse else bool(cond_features(cfg))
        arg = torch.cat((z1, y_onehot), 1) if use_y else z1
        return coupling_net(sd, pfx + "block.", arg, is_1d)

    if coupling == "made":
        return made_step(sd, cfg, pfx, x, logdet, y_onehot, reverse)

    if not reverse:
        z, logdet = actnorm(sd[pfx + "actnorm.bias"], sd[pfx + "actnorm.logs"], x, logdet, False)
        z, logdet = permutation(z, logdet, False)
        c = z.shape[1]
        z1, z2 = z[:, : c // 2], z[:, c // 2 :]
        h = net(z1)
        if coupling == "additive":
            z2 = z2 + h
        elif coupling == "affine":
            shift, scale = h[:, 0::2], torch.sigmoid(h[:, 1::2] + 2.0)
            z2 = (z2 + shift) * scale
            logdet = torch.log(scale).sum(red) + logdet
        else:
            raise NameError(f"Unknown coupling type: {coupling}")
        return torch.cat((z1, z2), 1), logdet

    c = x.shape[1]
    z1, z2 = x[:, : c // 2], x[:, c // 2 :]
    h = net(z1)
    if coupling == "additive":
        z2 = z2 - h
    elif coupling == "affine":
        shift, scale = h[:, 0::2], torch.sigmoid(h[:, 1::2] + 2.0)
        z2 = z2 / scale - shift
        logdet = -torch.log(scale).sum(red) + logdet
    else:
        raise NameError(f"Unknown coupling type: {coupling}")
    z = torch.cat((z1, z2), 1)
    z, logdet = permutation(z, logdet, True)
    z, logdet = actnorm(sd[pfx + "actnorm.bias"], sd[pfx + "actnorm.logs"], z, logdet, True)
    return z, logdet


def made_masks(d, hidden, ncond=0):
    """Autoregressive masks (Germain et al. 2015, MADE, eq. 12-13; MAF, Papamakarios et al. 2017, section 3.1) for an MLP of
    get_block_1d's shape: input degrees 1..d (condition features: 0), hidden degrees cycling over 1..d-1, outputs
    (mu_i, alpha_i) interleaved with degree i.  PARITY UNPINNED: the reference repository contains no MADE / MAF code
    (SURVEY.md section 0 F1); this restates the published algorithm, it is not pinned to any reference output."""
    deg_in = torch.cat([torch.arange(1, d + 1), torch.zeros(ncond, dtype=torch.long)])
    deg_h = torch.arange(hidden) % max(d - 1, 1) + 1
    deg_out = torch.arange(1, d + 1).repeat_interleave(2)
    masks = [(deg_h[:, None] >= deg_in[None, :])]
    masks += [(deg_h[:, None] >= deg_h[None, :]) for _ in range(4)]
    masks.append(deg_out[:, None] > deg_h[None, :])
    return masks


def made_net(sd, pfx, x, masks):
    h = x
    for k, i in enumerate((0, 2, 4, 6)):
        w = sd[f"{pfx}{i}.weight"]
        h = torch.relu(F.linear(h, w * masks[k].to(w.dtype), sd[f"{pfx}{i}.bias"]))
    w = sd[pfx + "8.weight"]
    h = torch.tanh(F.linear(h, w * masks[4].to(w.dtype), sd[pfx + "8.bias"]))
    w = sd[pfx + "10.weight"]
    return F.linear(h, w * masks[5].to(w.dtype), sd[pfx + "10.bias"])


def made_step(sd, cfg, pfx, x, logdet, y_onehot, reverse):
    """One MAF block: ActNorm1d -> reversal of the feature order -> u_i = (y_i - mu_i(y_<i)) exp(-alpha_i(y_<i)),
    log|det| = -sum_i alpha_i (Papamakarios et al. 2017, eq. 3-5); the inverse fixes one feature per pass (eq. 2).
    PARITY UNPINNED (see made_masks)."""
    d = x.shape[1]
    ncond = cond_features(cfg)
    hidden = sd[pfx + "block.0.weight"].shape[0]
    masks = made_masks(d, hidden, ncond)
    rev_idx = torch.arange(d - 1, -1, -1)

    def net(y):
        arg = torch.cat((y, y_onehot), 1) if ncond else y
        h = made_net(sd, pfx + "block.", arg, masks)
        return h[:, 0::2], h[:, 1::2]

    if not reverse:
        z, logdet = actnorm(sd[pfx + "actnorm.bias"], sd[pfx + "actnorm.logs"], x, logdet, False)
        y = z[:, rev_idx]
        mu, alpha = net(y)
        return (y - mu) * torch.exp(-alpha), logdet - alpha.sum(1)
    u = x
    y = torch.zeros_like(u)
    for i in range(d):
        mu, alpha = net(y)
        y = torch.cat((y[:, :i], (u[:, i] * torch.exp(alpha[:, i]) + mu[:, i])[:, None], y[:, i + 1:]), 1)
    mu, alpha = net(y)
    logdet = logdet + alpha.sum(1)
    z = y[:, rev_idx]
    return actnorm(sd[pfx + "actnorm.bias"], sd[pfx + "actnorm.logs"], z, logdet, True)


def split2d(sd, pfx, x, logdet, reverse, temperature=None, eps=None):
    """Split2d (models/layers.py:293-313).  In reverse, z2 = mean + exp(logs)*T*eps
    (what torch.normal(mean, exp(logs)*T) draws, with the N(0,1) variate made explicit).  torch.normal
    (models/layers.py:26-29) is not differentiable, so the draw is detached: a reverse pass under autograd
    sends no gradient into the prior conv."""
    if not reverse:
        c = x.shape[1]
        z1, z2 = x[:, : c // 2], x[:, c // 2 :]
        h = conv_zeros(sd, pfx + "conv.", z1)
        return z1, gaussian_logp(h[:, 0::2], h[:, 1::2], z2) + logdet
    h = conv_zeros(sd, pfx + "conv.", x)
    mean, logs = h[:, 0::2], h[:, 1::2]
    if eps is None:
        eps = torch.zeros_like(mean)
    z2 = (mean + torch.exp(logs) * temperature * eps).detach()
    return torch.cat((x, z2), 1), logdet


# --------------------------------------------------------------------------------------
# model level (models/flows.py:367-433, models/kd_flows.py:42-152)
# --------------------------------------------------------------------------------------
def prior(sd, cfg, batch, y_onehot=None):
    """Glow.prior (models/flows.py:367-391): returns (mean, logs) of the top latent."""
    is_1d = bool(cfg.get("is_1d", False))
    h = sd["prior_h"]
    h = h.repeat(batch, 1) if is_1d else h.repeat(batch, 1, 1, 1)
    ch = h.shape[1]
    if cfg.get("learn_top", False):
        h = linear_zeros(sd, "learn_top_fn.", h) if is_1d else conv_zeros(sd, "learn_top_fn.", h)
    if cfg.get("y_condition", False):
        assert y_onehot is not None
        yp = linear_zeros(sd, "project_ycond.", y_onehot)
        h = h + (yp.view(batch, ch) if is_1d else yp.view(batch, ch, 1, 1))
    return h[:, : ch // 2], h[:, ch // 2 :]


def encode(sd, cfg, x, logdet, y_onehot=None, perms=None):
    """FlowNetGetAllOutputs.encode (models/kd_flows.py:42-53): all layer outputs + logdet."""
    outs = []
    z = x
    for i, (kind, _) in enumerate(layer_schedule(cfg)):
        pfx = f"flow.layers.{i}."
        if kind == "squeeze":
            z = squeeze2(z)
        elif kind == "step":
            z, logdet = flow_step(sd, cfg, pfx, z, logdet, y_onehot, False,
                                  None if perms is None else perms.get(i))
        else:
            z, logdet = split2d(sd, pfx, z, logdet, False)
        outs.append(z)
    return outs, logdet


def decode(sd, cfg, z, y_onehot=None, temperature=None, eps=None, perms=None):
    """FlowNetGetAllOutputs.decode (models/kd_flows.py:55-73).  `eps` maps layer index ->
    N(0,1) tensor for that Split2d (None -> zeros)."""
    outs = []
    sched = layer_schedule(cfg)
    for i in reversed(range(len(sched))):
        kind = sched[i][0]
        pfx = f"flow.layers.{i}."
        if kind == "squeeze":
            z = unsqueeze2(z)
        elif kind == "step":
            z, _ = flow_step(sd, cfg, pfx, z, 0.0, y_onehot, True,
                             None if perms is None else perms.get(i))
        else:
            z, _ = split2d(sd, pfx, z, 0.0, True, temperature,
                           None if eps is None else eps.get(i))
        outs.append(z)
    return outs


def dequant_logdet(x):
    """logdet0 of uniform_binning_correction (models/utils.py:26-41)."""
    b, c, h, w = x.shape
    return torch.full((b,), -math.log(256.0) * c * h * w, dtype=x.dtype)


def glow_forward(sd, cfg, x, y_onehot=None, noise=None, perms=None):
    """GlowGetAllOutputs.normal_flow (models/kd_flows.py:121-152).

    2-D: `x` is mutated in place (x += noise) like the reference (models/utils.py:38);
    `noise` must be given explicitly (the U(0,1/256) draw), None means no noise is added.
    Returns (all layer outputs, bpd, y_logits)."""
    is_1d = bool(cfg.get("is_1d", False))
    if is_1d:
        b, c = x.shape
        logdet = torch.zeros(b, dtype=x.dtype)
    else:
        b, c, h, w = x.shape
        if noise is not None:
            x += noise
        logdet = dequant_logdet(x)
    outs, objective = encode(sd, cfg, x, logdet, y_onehot, perms)
    last = outs[-1]
    mean, logs = prior(sd, cfg, b, y_onehot)
    objective = objective + gaussian_logp(mean, logs, last)
    y_logits = None
    if cfg.get("y_condition", False):
        pooled = last if is_1d else last.mean(dim=[2, 3])
        y_logits = linear_zeros(sd, "project_class.", pooled)
    bpd = -objective if is_1d else -objective / (math.log(2.0) * c * h * w)
    return outs, bpd, y_logits


def glow_reverse(sd, cfg, z=None, y_onehot=None, temperature=None, eps=None, eps_top=None,
                 perms=None):
    """Glow.reverse_flow (models/flows.py:428-433) with explicit N(0,1) variates."""
    if z is None:
        batch = y_onehot.shape[0] if y_onehot is not None else 32
        mean, logs = prior(sd, cfg, batch, y_onehot)
        if eps_top is None:
            eps_top = torch.zeros_like(mean)
        z = mean + torch.exp(logs) * temperature * eps_top
    return decode(sd, cfg, z, y_onehot, temperature, eps, perms)


# --------------------------------------------------------------------------------------
# distillation step (pl_module.py:81-110, 198-320, 350-354; train.py:46)
# --------------------------------------------------------------------------------------
def kd_indices(student_cfg, teacher_cfg):
    """NFModel._get_kd_indices (pl_module.py:81-110)."""
    s_1d = bool(student_cfg.get("is_1d", False))

    def pick(cfg, mult):
        sched = layer_schedule(cfg)
        return [
            i
            for i, (kind, _) in enumerate(sched)
            if kind == "squeeze" or (s_1d and (i + 1) % mult == 0) or i + 1 == len(sched)
        ]

    return pick(student_cfg, 2), pick(teacher_cfg, 4)


def distill_losses(student_z, student_nll, teacher_z, s_idx, t_idx, w_nll, w_kd):
    """NFModel.loss (pl_module.py:257-320) with perceptual weight 0.  The hint lists are ZIPPED (the shorter one
    decides how many pairs there are) while the divisor is len(student_kd_indices) (:264-282)."""
    kd = None
    if w_kd > 0:
        for si, ti in zip(s_idx, t_idx):
            d = ((student_z[si] - teacher_z[ti]) ** 2).flatten(1).mean(1)
            kd = d if kd is None else kd + d
        kd = kd / (len(s_idx) if s_idx else 1)
    else:
        kd = torch.zeros((), dtype=student_nll.dtype)
    res = w_nll * student_nll + w_kd * kd
    return {"nll": student_nll.mean(), "kd": kd.mean(), "result_loss": res.mean(), "per_sample": res}


def clip_grad_norm(grads: List[Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ semantics (PL gradient_clip_val=30, train.py:46)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (non-amsgrad) single step, in place (pl_module.py:350-354)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    for k, (p, g) in enumerate(zip(params, grads)):
        if weight_decay:
            g = g + weight_decay * p
        m = state.setdefault(("m", k), torch.zeros_like(p))
        v = state.setdefault(("v", k), torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


TRAINABLE_EXCLUDE = ("prior_h", ".p", ".sign_s")


def is_buffer(key: str) -> bool:
    return key == "prior_h" or key.endswith("invconv.p") or key.endswith("invconv.sign_s")


def perceptual_l1(student_sd, student_cfg, teacher_sd, teacher_cfg, latent, y_onehot=None):
    """The perceptual branch for tabular data (pl_module.py:229-243, 283-299 with nn.L1Loss(reduction='none')):
    both models decode the SAME latent at temperature 0.7 (the student with gradients); per-sample mean |dx|,
    NaN rows zeroed.  `latent` = gaussian_sample(student prior, 1), drawn by the caller."""
    s_x = glow_reverse(student_sd, student_cfg, latent, y_onehot, temperature=0.7)[-1]
    with torch.no_grad():
        t_x = glow_reverse(teacher_sd, teacher_cfg, latent, y_onehot, temperature=0.7)[-1]
    per = (s_x - t_x).abs().flatten(1).mean(1)
    return torch.where(torch.isnan(per), torch.zeros_like(per), per), s_x, t_x


def distill_step(student_sd, student_cfg, teacher_sd, teacher_cfg, x, noise_s, noise_t,
                 opt_state, lr=5e-4, w_nll=0.9, w_kd=0.1, clip=30.0, w_perc=0.0, latent=None,
                 y_onehot=None, weights=None, keep_grads=False):
    """One training step as the reference runs it: student fwd (grad), teacher fwd on the
    SAME (already noised, in-place) x (no grad), MSE hints, [perceptual branch], [per-sample weights,
    pl_module.py:309-310], clip, Adam on the student.  `y_onehot` goes to both forwards, the prior and both
    reverse passes when the student is conditional (pl_module.py:216-243).
    Returns dict of scalars; mutates student_sd and x."""
    names = [k for k in student_sd if not is_buffer(k)]
    for k in names:
        student_sd[k].requires_grad_(True)
        student_sd[k].grad = None
    s_idx, t_idx = kd_indices(student_cfg, teacher_cfg)
    s_z, s_nll, _ = glow_forward(student_sd, student_cfg, x, y_onehot, noise_s)
    with torch.no_grad():
        t_z, _, _ = glow_forward(teacher_sd, teacher_cfg, x, y_onehot, noise_t)
    out = distill_losses(s_z, s_nll, t_z, s_idx, t_idx, w_nll, w_kd)
    per_sample = out.pop("per_sample")
    if w_perc > 0:
        per, _, _ = perceptual_l1(student_sd, student_cfg, teacher_sd, teacher_cfg, latent, y_onehot)
        out["perceptual"] = per.mean()
        per_sample = per_sample + w_perc * per
    if weights is not None:
        per_sample = per_sample * weights
    out["result_loss"] = per_sample.mean()
    out["result_loss"].backward()
    with torch.no_grad():
        params = [student_sd[k] for k in names]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        out["grad_norm"] = clip_grad_norm(grads, clip)
        if keep_grads:
            out["grads"] = {k: g.clone() for k, g in zip(names, grads)}
        for p in params:
            p.requires_grad_(False)
        adam_step(params, grads, opt_state, lr)
    return {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in out.items()}


# --------------------------------------------------------------------------------------
# permutation inheritance (models/kd_flows.py:162-179) and the image pipeline (data/src/utils.py:7-25)
# --------------------------------------------------------------------------------------
def inherit_permutation_matrix(student_sd, teacher_sd, teacher_cfg, s_idx, t_idx):
    """models/kd_flows.py:162-179: at every teacher hint layer the student's hint layer gets
    p = (product of the teacher's p since the previous hint layer) @ (the hint layer's p).
    With nothing accumulated the reference computes `None @ p` -> TypeError; kept."""
    k, acc = 0, None
    for t_id, (kind, _) in enumerate(layer_schedule(teacher_cfg)):
        p = teacher_sd.get(f"flow.layers.{t_id}.invconv.p")
        if t_id == t_idx[k]:
            if acc is None:
                raise TypeError("unsupported operand type(s) for @: 'NoneType' and 'Tensor'")
            student_sd[f"flow.layers.{s_idx[k]}.invconv.p"] = acc @ p
            k += 1
            acc = None
        elif kind == "step":
            acc = p if acc is None else acc @ p


def preprocess(x):
    """data/src/utils.py:7-18 with n_bits = 8: x in [0,1] (ToTensor) -> x*255/256 - 0.5."""
    return x * 255 / 256 - 0.5


def postprocess(x):
    """data/src/utils.py:21-25."""
    x = torch.clamp(x, -0.5, 0.5) + 0.5
    return torch.clamp(x * 255, 0, 255).byte()
