#!/usr/bin/env python3
"""Generate golden vectors from the reference's own code (runs ONLY in the build container).

The reference's hot path (`models/layers.py`, `models/flows.py`, `models/kd_flows.py`,
`models/utils.py`) is imported from /root/reference under a stub parent package, because
`models/__init__.py:4` imports torchvision (absent).  Nothing of the reference is copied:
the outputs below are *data* -- inputs, parameters and the tensors the reference computed.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

`pl_module.py` cannot be imported (pytorch_lightning/hydra/neptune absent), so the
distillation-step case drives the reference *models* with torch's own Adam and
clip_grad_norm_ following pl_module.py:198-320,350-354 and train.py:46.
"""
import importlib
import json
import math
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from golden_util import tensor_digest  # noqa: E402
from param_util import randomize_state_dict, shapes_of, synthetic_images  # noqa: E402

REF = "/root/reference/models"


def import_reference():
    pkg = types.ModuleType("models")
    pkg.__path__ = [REF]
    sys.modules["models"] = pkg
    layers = importlib.import_module("models.layers")
    flows = importlib.import_module("models.flows")
    kd = importlib.import_module("models.kd_flows")
    return layers, flows, kd


layers, flows, kd = import_reference()


def base_cfg(**kw):
    cfg = dict(image_shape=[8, 8, 3], hidden_channels=16, K=2, L=2, actnorm_scale=1.0,
               flow_permutation="invconv", flow_coupling="affine", LU_decomposed=True,
               y_classes=10, learn_top=False, y_condition=False, is_1d=False)
    cfg.update(kw)
    return cfg


def build(cfg, seed):
    torch.manual_seed(1234)
    model = kd.create_glow_model(dict(cfg))
    sd = randomize_state_dict(shapes_of(model.state_dict()), seed)
    model.load_state_dict(sd)
    return model, sd


def perms_of(model):
    out = {}
    for i, layer in enumerate(model.flow.layers):
        for name in ("shuffle", "reverse"):
            if isinstance(getattr(layer, name, None), layers.Permute2d):
                out[i] = getattr(layer, name).indices.clone()
    return out


def save(name, cfg, arrays):
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
              for k, v in arrays.items()}
    arrays["cfg"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def model_case(name, cfg, batch, seed, noise_seed=99):
    model, sd = build(cfg, seed)
    is_1d = cfg["is_1d"]
    g = torch.Generator().manual_seed(seed + 17)
    arrays = {f"sd/{k}": v for k, v in sd.items()}
    y = None
    if cfg["y_condition"]:
        y = torch.zeros(batch, cfg["y_classes"])
        y[torch.arange(batch), torch.randint(0, cfg["y_classes"], (batch,), generator=g)] = 1.0
        arrays["y"] = y
    if is_1d:
        x = torch.randn(batch, cfg["image_shape"][0], generator=g)
        arrays["x"] = x.clone()
    else:
        h, w, c = cfg["image_shape"]
        x, _ = synthetic_images(batch, (c, h, w), seed=seed + 5)
        arrays["x"] = x.clone()
        torch.manual_seed(noise_seed)
        noise = torch.zeros_like(x).uniform_(0, 1.0 / 256)
        arrays["noise"] = noise
        torch.manual_seed(noise_seed)  # the reference now draws the same noise
    for i, t in perms_of(model).items():
        arrays[f"perm/{i}"] = t
    model.train()
    x_in = x.clone()
    outs, bpd, y_logits = model(x_in, y)
    if not is_1d:
        assert torch.equal(x_in, arrays["x"] + arrays["noise"]), "noise replay mismatch"
    loss = bpd.mean() + (0.1 * (y_logits ** 2).sum() if y_logits is not None else 0.0)
    loss.backward()
    for i, o in enumerate(outs):
        arrays[f"out/{i}"] = o
    arrays["bpd"] = bpd
    if y_logits is not None:
        arrays["y_logits"] = y_logits
    for k, p in model.named_parameters():
        arrays[f"grad/{k}"] = p.grad if p.grad is not None else torch.zeros_like(p)
    # reverse pass from the top latent, temperature 0 (Split2d then returns its mean:
    # torch.normal(mean, 0) == mean), so no RNG is involved.
    with torch.no_grad():
        rev = model(z=outs[-1].detach(), y_onehot=y, temperature=0.0, reverse=True)
        for i, o in enumerate(rev):
            arrays[f"rev/{i}"] = o
        mean, logs = model.prior(outs[-1], y)
        arrays["prior_mean"], arrays["prior_logs"] = mean, logs
    # fp64 run of the same reference code: error budget for the fp32 comparison
    m64, _ = build(cfg, seed)
    m64 = m64.double()
    with torch.no_grad():
        xin64 = (arrays["x"] + arrays["noise"]).double() if not is_1d else arrays["x"].double()
        ld0 = (torch.zeros(batch, dtype=torch.float64) if is_1d else
               torch.full((batch,), -math.log(256.0) * xin64[0].numel(), dtype=torch.float64))
        o64, obj64 = m64.flow(xin64, y_onehot=None if y is None else y.double(), logdet=ld0,
                              reverse=False)
        arrays["logdet64"] = obj64
        arrays["last64"] = o64[-1]
    save(name, cfg, arrays)


def reverse_grad_case(name, cfg, batch, seed):
    """Reverse (sampling) pass WITH gradients, as the perceptual branch of the tabular distillation loss uses it
    (pl_module.py:229-243, 283-299): x = model(z=latent, temperature=0.7, reverse=True)[-1],
    loss = L1(x, target).mean(dim=1).mean(); gradients wrt every parameter and wrt the latent."""
    model, sd = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 23)
    d = cfg["image_shape"][0]
    arrays = {f"sd/{k}": v for k, v in sd.items()}
    y = None
    if cfg["y_condition"]:
        y = torch.zeros(batch, cfg["y_classes"])
        y[torch.arange(batch), torch.randint(0, cfg["y_classes"], (batch,), generator=g)] = 1.0
        arrays["y"] = y
    z = torch.randn(batch, d, generator=g).requires_grad_(True)
    target = torch.randn(batch, d, generator=g)
    arrays["z"], arrays["target"] = z.detach().clone(), target
    model.train()
    outs = model(z=z, y_onehot=y, temperature=0.7, reverse=True)
    x = outs[-1]
    per = torch.nn.L1Loss(reduction="none")(x, target).mean(dim=1)
    loss = per.mean()
    loss.backward()
    for i, o in enumerate(outs):
        arrays[f"rev/{i}"] = o
    arrays["loss"] = loss
    arrays["grad_z"] = z.grad
    for k, p in model.named_parameters():
        arrays[f"grad/{k}"] = p.grad if p.grad is not None else torch.zeros_like(p)
    save(name, cfg, arrays)


def reverse_grad_2d_case(name, cfg, batch, seed):
    """The reverse pass WITH gradients on an IMAGE flow (pl_module.py:229-243 runs it for any student whose
    perceptual_weight > 0): x = model(z=latent, temperature=0.7, reverse=True)[-1], loss = L1(x, target) averaged over
    every dim but the batch, then over the batch (pl_module.py:288-291).  Split2d draws with torch.normal; the N(0,1)
    variate of each draw is recovered from the outputs ((z2 - mean) / (exp(logs) T)) and stored, so that the draw can be
    reproduced."""
    model, sd = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 23)
    arrays = {f"sd/{k}": v for k, v in sd.items()}
    H, W, Cimg = cfg["image_shape"]
    with torch.no_grad():
        x0 = torch.rand(batch, Cimg, H, W, generator=g)
        top_shape = model.flow(x0, logdet=torch.zeros(batch), reverse=False)[0][-1].shape
    z = torch.randn(*top_shape, generator=g).requires_grad_(True)
    target = torch.rand(batch, Cimg, H, W, generator=g) - 0.5
    arrays["z"], arrays["target"] = z.detach().clone(), target
    for i, t in perms_of(model).items():
        arrays[f"perm/{i}"] = t
    model.train()
    torch.manual_seed(seed + 77)
    outs = model(z=z, y_onehot=None, temperature=0.7, reverse=True)
    x = outs[-1]
    per = torch.nn.L1Loss(reduction="none")(x, target)
    per = per.mean(dim=list(range(len(per.shape)))[1:])
    loss = per.mean()
    loss.backward()
    n_layers = len(model.flow.layers)
    with torch.no_grad():
        for n, layer in enumerate(reversed(list(model.flow.layers))):
            if isinstance(layer, layers.Split2d):
                o = outs[n]
                c = o.shape[1] // 2
                mean, logs = layer.split2d_prior(o[:, :c])
                arrays[f"eps/{n_layers - 1 - n}"] = (o[:, c:] - mean) / (torch.exp(logs) * 0.7)
    for i, o in enumerate(outs):
        arrays[f"rev/{i}"] = o
    arrays["loss"] = loss
    arrays["grad_z"] = z.grad
    for k, p in model.named_parameters():
        arrays[f"grad/{k}"] = p.grad if p.grad is not None else torch.zeros_like(p)
    save(name, cfg, arrays)


def reverse_grad_2d_cases():
    reverse_grad_2d_case("glow8_revgrad", base_cfg(), 2, 131)
    reverse_grad_2d_case("glow16_additive_nolu_revgrad",
                         base_cfg(image_shape=[16, 16, 3], hidden_channels=24, K=2, L=3, flow_coupling="additive",
                                  LU_decomposed=False), 3, 132)
    reverse_grad_2d_case("glow8_shuffle_revgrad", base_cfg(flow_permutation="shuffle"), 2, 133)


def primitive_case():
    """Per-primitive vectors."""
    g = torch.Generator().manual_seed(3)
    arrays = {}
    x = torch.randn(2, 3, 4, 6, generator=g)
    arrays["squeeze/x"] = x
    arrays["squeeze/y"] = layers.squeeze2d(x, 2)
    arrays["unsqueeze/y"] = layers.unsqueeze2d(layers.squeeze2d(x, 2), 2)
    mean, logs, z = (torch.randn(3, 4, 2, 2, generator=g) for _ in range(3))
    arrays["gauss/mean"], arrays["gauss/logs"], arrays["gauss/x"] = mean, logs * 0.3, z
    arrays["gauss/lp"] = layers.gaussian_likelihood(mean, logs * 0.3, z)
    for tag, is_1d, shape in (("an2d", False, (3, 6, 4, 4)), ("an1d", True, (5, 7))):
        c = shape[1]
        an = (layers.ActNorm1d if is_1d else layers.ActNorm2d)(c)
        an.inited = True
        an.bias.data = torch.randn(an.bias.shape, generator=g) * 0.3   # seeded: the fixture is reproducible
        an.logs.data = torch.randn(an.logs.shape, generator=g) * 0.3
        xx = torch.randn(*shape, generator=g)
        ld = torch.randn(shape[0], generator=g)
        y, ldo = an(xx, ld, reverse=False)
        yi, ldi = an(xx, ld, reverse=True)
        arrays.update({f"{tag}/bias": an.bias.data, f"{tag}/logs": an.logs.data, f"{tag}/x": xx,
                       f"{tag}/ld": ld, f"{tag}/y": y, f"{tag}/ldo": ldo, f"{tag}/yi": yi,
                       f"{tag}/ldi": ldi})
    for tag, is_1d, lu, shape in (("ic2d", False, True, (2, 8, 3, 3)),
                                  ("ic1d", True, True, (4, 7)),
                                  ("ic2d_nolu", False, False, (2, 6, 2, 2))):
        c = shape[1]
        torch.manual_seed(11)
        ic = layers.InvertibleConv1x1(c, LU_decomposed=lu, is_1d=is_1d)
        sd = randomize_state_dict({f"invconv.{k}": v.shape for k, v in ic.state_dict().items()}, 21)
        ic.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items()})
        xx = torch.randn(*shape, generator=g)
        ld = torch.randn(shape[0], generator=g)
        y, ldo = ic(xx, ld, reverse=False)
        yi, ldi = ic(xx, ld, reverse=True)
        for k, v in sd.items():
            arrays[f"{tag}/sd/{k}"] = v
        arrays.update({f"{tag}/x": xx, f"{tag}/ld": ld, f"{tag}/y": y, f"{tag}/ldo": ldo,
                       f"{tag}/yi": yi, f"{tag}/ldi": ldi})
    # Split2d forward and reverse at temperature 0
    sp = layers.Split2d(8)
    sd = randomize_state_dict({f"conv.{k}": v.shape for k, v in sp.conv.state_dict().items()}, 31)
    sp.conv.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items()})
    xx = torch.randn(2, 8, 4, 4, generator=g)
    ld = torch.randn(2, generator=g)
    z1, ldo = sp(xx, ld, reverse=False)
    zr, _ = sp(z1, 0.0, reverse=True, temperature=0.0)
    for k, v in sd.items():
        arrays[f"split/sd/{k}"] = v
    arrays.update({"split/x": xx, "split/ld": ld, "split/z1": z1, "split/ldo": ldo, "split/rev": zr})
    save("primitives", {}, arrays)


def distill_case():
    """Two optimisation steps of the distillation training_step on a tiny model pair."""
    s_cfg = base_cfg(K=1, hidden_channels=16)
    t_cfg = base_cfg(K=2, hidden_channels=16)
    student, s_sd = build(s_cfg, 41)
    teacher, t_sd = build(t_cfg, 43)
    teacher.eval()
    s_idx = [i for i, l in enumerate(student.flow.layers)
             if isinstance(l, layers.SqueezeLayer) or i + 1 == len(student.flow.layers)]
    t_idx = [i for i, l in enumerate(teacher.flow.layers)
             if isinstance(l, layers.SqueezeLayer) or i + 1 == len(teacher.flow.layers)]
    opt = torch.optim.Adam(student.parameters(), lr=5e-4, weight_decay=0)
    arrays = {f"s_sd/{k}": v for k, v in s_sd.items()}
    arrays.update({f"t_sd/{k}": v for k, v in t_sd.items()})
    arrays["s_idx"], arrays["t_idx"] = np.array(s_idx), np.array(t_idx)
    for step in range(2):
        x, _ = synthetic_images(4, (3, 8, 8), seed=50 + step)
        arrays[f"step{step}/x"] = x.clone()
        torch.manual_seed(70 + step)
        n_s = torch.zeros_like(x).uniform_(0, 1.0 / 256)
        n_t = torch.zeros_like(x).uniform_(0, 1.0 / 256)
        arrays[f"step{step}/noise_s"], arrays[f"step{step}/noise_t"] = n_s, n_t
        torch.manual_seed(70 + step)
        opt.zero_grad()
        s_z, s_nll, _ = student(x, None)               # mutates x in place
        with torch.no_grad():
            t_z, _, _ = teacher(x, None)               # sees student noise + its own
        assert torch.allclose(x, arrays[f"step{step}/x"] + n_s + n_t, atol=0, rtol=0)
        kd_val = None
        for si, ti in zip(s_idx, t_idx):
            d = torch.nn.functional.mse_loss(s_z[si], t_z[ti], reduction="none")
            d = d.mean(dim=list(range(d.dim()))[1:])
            kd_val = d if kd_val is None else kd_val + d
        kd_val = kd_val / len(s_idx)
        res = (0.9 * s_nll + 0.1 * kd_val).mean()
        res.backward()
        gn = torch.nn.utils.clip_grad_norm_(student.parameters(), 30.0)
        for k, p in student.named_parameters():
            arrays[f"step{step}/grad/{k}"] = p.grad.clone()
        opt.step()
        arrays[f"step{step}/nll"] = s_nll.mean()
        arrays[f"step{step}/kd"] = kd_val.mean()
        arrays[f"step{step}/loss"] = res
        arrays[f"step{step}/grad_norm"] = gn
        for k, v in student.state_dict().items():
            arrays[f"step{step}/s_sd_after/{k}"] = v.clone()
    save("distill8", {"student": s_cfg, "teacher": t_cfg}, arrays)


def distill_tab_case():
    """Two optimisation steps of the TABULAR distillation step (conf/training/tabular.yaml: nll 0.85, kd 0.05 (mse),
    perceptual 0.1 (l1)): NFModel.forward + loss (pl_module.py:198-320) restated on the reference's models, including
    the perceptual branch -- reverse passes of student (with gradients) and teacher from a latent drawn from the
    student's prior -- then clip 30 + Adam.  The drawn latents are stored (the product takes them as an argument)."""
    tab = dict(L=1, is_1d=True, y_classes=0, hidden_channels=32, image_shape=[6])
    s_cfg, t_cfg = base_cfg(K=3, **tab), base_cfg(K=5, **tab)
    student, s_sd = build(s_cfg, 61)
    teacher, t_sd = build(t_cfg, 63)
    teacher.eval()
    s_idx, t_idx = [1, 2], [3, 4]      # NFModel._get_kd_indices for 1-D K=3 / K=5 (pl_module.py:81-110)
    opt = torch.optim.Adam(student.parameters(), lr=5e-4, weight_decay=0)
    arrays = {f"s_sd/{k}": v for k, v in s_sd.items()}
    arrays.update({f"t_sd/{k}": v for k, v in t_sd.items()})
    arrays["s_idx"], arrays["t_idx"] = np.array(s_idx), np.array(t_idx)
    l1 = torch.nn.L1Loss(reduction="none")
    for step in range(2):
        g = torch.Generator().manual_seed(80 + step)
        x = torch.randn(16, 6, generator=g)
        arrays[f"step{step}/x"] = x.clone()
        opt.zero_grad()
        s_z, s_nll, _ = student(x, None)
        with torch.no_grad():
            t_z, _, _ = teacher(x, None)
        kd_val = None
        for si, ti in zip(s_idx, t_idx):
            d = torch.nn.functional.mse_loss(s_z[si], t_z[ti], reduction="none")
            d = d.mean(dim=list(range(d.dim()))[1:])
            kd_val = d if kd_val is None else kd_val + d
        kd_val = kd_val / len(s_idx)
        mean, logs = student.prior(x, y_onehot=None)
        torch.manual_seed(90 + step)
        latent = layers.gaussian_sample(mean, logs, 1)
        arrays[f"step{step}/latent"] = latent.clone()
        student_x = student(z=latent, temperature=0.7, reverse=True, y_onehot=None)[-1]
        with torch.no_grad():
            teacher_x = teacher(z=latent, temperature=0.7, reverse=True, y_onehot=None)[-1]
        perc = l1(student_x, teacher_x)
        perc = perc.mean(dim=list(range(perc.dim()))[1:])
        perc[(torch.isnan(perc)).nonzero(as_tuple=True)[0]] = 0
        res = (0.85 * s_nll + 0.05 * kd_val + 0.1 * perc).mean()
        res.backward()
        gn = torch.nn.utils.clip_grad_norm_(student.parameters(), 30.0)
        for k, p in student.named_parameters():
            arrays[f"step{step}/grad/{k}"] = p.grad.clone()
        opt.step()
        arrays[f"step{step}/nll"], arrays[f"step{step}/kd"] = s_nll.mean(), kd_val.mean()
        arrays[f"step{step}/perceptual"], arrays[f"step{step}/loss"] = perc.mean(), res
        arrays[f"step{step}/grad_norm"] = gn
        arrays[f"step{step}/student_x"], arrays[f"step{step}/teacher_x"] = student_x, teacher_x
        for k, v in student.state_dict().items():
            arrays[f"step{step}/s_sd_after/{k}"] = v.clone()
    save("distill_tab6", {"student": s_cfg, "teacher": t_cfg}, arrays)


def checksum_case(name, cfg, batch, seed):
    """Full-size models: parameters are rebuilt from (shapes, seed) by the tests, only
    per-layer checksums and bpd are stored."""
    model, _ = build(cfg, seed)
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(batch, (c, h, w), seed=seed + 5)
    xin = x + noise
    with torch.no_grad():
        ld0 = torch.full((batch,), -math.log(256.0) * c * h * w)
        outs, obj = model.flow(xin.clone(), y_onehot=None, logdet=ld0, reverse=False)
        mean, logs = model.prior(xin, None)
        obj = obj + layers.gaussian_likelihood(mean, logs, outs[-1])
        bpd = -obj / (math.log(2.0) * c * h * w)
    arrays = {"bpd": bpd, "objective": obj,
              "out_mean": torch.stack([o.double().mean() for o in outs]),
              "out_absmean": torch.stack([o.double().abs().mean() for o in outs]),
              "seed": np.array(seed), "batch": np.array(batch)}
    save(name, cfg, arrays)


def reverse_grad_cases():
    tab = dict(L=1, is_1d=True, y_classes=0)
    reverse_grad_case("tab6_revgrad", base_cfg(image_shape=[6], K=2, **tab), 8, 121)
    reverse_grad_case("tab63_revgrad", base_cfg(image_shape=[63], K=2, **tab), 8, 122)
    reverse_grad_case("tab6_k5_h32_revgrad", base_cfg(image_shape=[6], K=5, hidden_channels=32, **tab), 16, 123)
    reverse_grad_case("tab6_additive_nolu_revgrad",
                      base_cfg(image_shape=[6], K=3, flow_coupling="additive", LU_decomposed=False, **tab), 8, 125)
    reverse_grad_case("tab5_cond_revgrad", base_cfg(image_shape=[5], K=2, L=1, is_1d=True, y_classes=3,
                                                    y_condition=True, learn_top=True), 8, 124)


# ------------------------------------------------------------------------------------------------
# round 2: full-size gradients / full-size distillation step, permutation inheritance, image pipeline,
# the RICH-style conditional distillation step with unequal hint lists
# ------------------------------------------------------------------------------------------------
def full_grads_case(name, cfg, batch, seed):
    """Full-size model (hidden 512): bpd.mean().backward() on the reference; parameters are rebuilt by the tests
    from (shapes, seed); stored: bpd and a digest of every parameter gradient."""
    model, _ = build(cfg, seed)
    model.train()
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(batch, (c, h, w), seed=seed + 5)
    xin = (x + noise).clone()
    ld0 = torch.full((batch,), -math.log(256.0) * c * h * w)
    outs, obj = model.flow(xin, y_onehot=None, logdet=ld0, reverse=False)
    mean, logs = model.prior(xin, None)
    obj = obj + layers.gaussian_likelihood(mean, logs, outs[-1])
    bpd = -obj / (math.log(2.0) * c * h * w)
    bpd.mean().backward()
    named = [(k, p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()]
    arrays = {"bpd": bpd, "seed": np.array(seed), "batch": np.array(batch),
              "n_params": np.array(len(named))}
    tensor_digest(arrays, "grad/", named)
    save(name, cfg, arrays)


def distill_full_case():
    """Two optimisation steps of the CIFAR distillation step at FULL size (teacher K=32 -> student K=8, hidden 512,
    B=4): pl_module.py:198-320 + :350-354 + train.py:46 on the reference's models, as distill_case does on toys.
    Parameters are rebuilt by the tests from (shapes, seed) -- bench.py's seeds 11 / 12."""
    cifar = dict(image_shape=[32, 32, 3], hidden_channels=512, L=3)
    s_cfg, t_cfg = base_cfg(K=8, **cifar), base_cfg(K=32, **cifar)
    student, _ = build(s_cfg, 11)
    teacher, _ = build(t_cfg, 12)
    teacher.eval()
    pick = lambda m: [i for i, l in enumerate(m.flow.layers)
                      if isinstance(l, layers.SqueezeLayer) or i + 1 == len(m.flow.layers)]
    s_idx, t_idx = pick(student), pick(teacher)
    opt = torch.optim.Adam(student.parameters(), lr=5e-4, weight_decay=0)
    arrays = {"s_idx": np.array(s_idx), "t_idx": np.array(t_idx), "s_seed": np.array(11), "t_seed": np.array(12),
              "batch": np.array(4)}
    for step in range(2):
        x, _ = synthetic_images(4, (3, 32, 32), seed=300 + step)
        g = torch.Generator().manual_seed(310 + step)
        n_s = torch.rand(x.shape, generator=g) / 256.0
        n_t = torch.rand(x.shape, generator=g) / 256.0
        arrays[f"step{step}/noise_seed"] = np.array(310 + step)
        opt.zero_grad()
        # explicit noise (the reference's own draw is replayed bit-exactly in distill_case; here the dequantised
        # input goes straight into .flow, exactly what normal_flow does after uniform_binning_correction)
        ld0 = torch.full((4,), -math.log(256.0) * 3 * 32 * 32)
        xs = x + n_s
        s_z, obj = student.flow(xs, y_onehot=None, logdet=ld0.clone(), reverse=False)
        mean, logs = student.prior(xs, None)
        obj = obj + layers.gaussian_likelihood(mean, logs, s_z[-1])
        s_nll = -obj / (math.log(2.0) * 3 * 32 * 32)
        with torch.no_grad():
            t_z, _ = teacher.flow(xs + n_t, y_onehot=None, logdet=ld0.clone(), reverse=False)
        kd_val = None
        for si, ti in zip(s_idx, t_idx):
            d = torch.nn.functional.mse_loss(s_z[si], t_z[ti], reduction="none")
            d = d.mean(dim=list(range(d.dim()))[1:])
            kd_val = d if kd_val is None else kd_val + d
        kd_val = kd_val / len(s_idx)
        res = (0.9 * s_nll + 0.1 * kd_val).mean()
        res.backward()
        gn = torch.nn.utils.clip_grad_norm_(student.parameters(), 30.0)
        tensor_digest(arrays, f"step{step}/grad/", [(k, p.grad) for k, p in student.named_parameters()])
        opt.step()
        arrays[f"step{step}/nll"], arrays[f"step{step}/kd"] = s_nll.mean(), kd_val.mean()
        arrays[f"step{step}/loss"], arrays[f"step{step}/grad_norm"] = res, gn
        arrays[f"step{step}/bpd"] = s_nll
        tensor_digest(arrays, f"step{step}/after/", list(student.named_parameters()))
    save("distill_full", {"student": s_cfg, "teacher": t_cfg}, arrays)


def inherit_case():
    """inherit_permutation_matrix (models/kd_flows.py:162-179, called at pl_module.py:71) on 1-D flows.
    Working geometry: teacher K=8 / student K=4 (hint layers [3,7] / [1,3]: FlowSteps lie between consecutive
    teacher hint layers and the last hint is the last layer).  Also recorded: what the reference does for
    K=5 / K=3 with [3,4] / [1,2] (adjacent hint layers -> `None @ p`)."""
    tab = dict(L=1, is_1d=True, y_classes=0, image_shape=[6])
    s_cfg, t_cfg = base_cfg(K=4, **tab), base_cfg(K=8, **tab)
    student, s_sd = build(s_cfg, 131)
    teacher, t_sd = build(t_cfg, 133)
    s_idx, t_idx = [1, 3], [3, 7]
    kd.inherit_permutation_matrix(student, teacher, s_idx, t_idx)
    arrays = {f"s_sd/{k}": v for k, v in s_sd.items()}
    arrays.update({f"t_sd/{k}": v for k, v in t_sd.items()})
    arrays["s_idx"], arrays["t_idx"] = np.array(s_idx), np.array(t_idx)
    for i in s_idx:
        arrays[f"p_after/{i}"] = student.flow.layers[i].invconv.p.clone()
    g = torch.Generator().manual_seed(135)
    x = torch.randn(8, 6, generator=g)
    arrays["x"] = x.clone()
    with torch.no_grad():
        outs, bpd, _ = student(x, None)
    arrays["bpd"] = bpd
    arrays["last"] = outs[-1]
    # the adjacent-hint geometry the tabular K=3 <- K=5 pair has
    s2, _ = build(base_cfg(K=3, **tab), 137)
    t2, _ = build(base_cfg(K=5, **tab), 139)
    try:
        kd.inherit_permutation_matrix(s2, t2, [1, 2], [3, 4])
        err = "none"
    except Exception as exc:  # noqa: BLE001
        err = type(exc).__name__
    arrays["adjacent_error"] = np.frombuffer(err.encode(), dtype=np.uint8)
    save("inherit_perm", {"student": s_cfg, "teacher": t_cfg}, arrays)


def image_pipeline_case():
    """data/src/utils.py:7-25 (`preprocess`, `postprocess`), imported from the reference file (it needs only torch).
    torchvision is absent, so ToTensor's `uint8 HWC -> float CHW / 255` in front of preprocess and
    RandomHorizontalFlip's mirror are written out here (data/src/cifar10.py:14-25 composes them in that order)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_data_utils", "/root/reference/data/src/utils.py")
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    g = torch.Generator().manual_seed(141)
    arrays = {}
    allv = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1)
    u = torch.randint(0, 256, (6, 32, 32, 3), dtype=torch.uint8, generator=g)
    flip = torch.rand(6, generator=g) < 0.5
    to_tensor = lambda t: t.permute(0, 3, 1, 2).to(torch.float32).div(255)
    arrays["all/u8"], arrays["all/x"] = allv, du.preprocess(to_tensor(allv))
    arrays["rand/u8"], arrays["rand/flip"] = u, flip
    arrays["rand/x"] = du.preprocess(to_tensor(u))
    uf = torch.where(flip.view(-1, 1, 1, 1), u.flip(2), u)       # mirror W of the HWC image, before ToTensor
    arrays["rand/x_flipped"] = du.preprocess(to_tensor(uf))
    x = torch.randn(5, 3, 16, 16, generator=g) * 0.4
    x[0, 0, 0, :8] = torch.tensor([-0.5, 0.5, -0.50001, 0.49999, 0.0, 1e9, -1e9, 0.5 - 1.0 / 512])
    x[0, 0, 1, :2] = torch.tensor([float("inf"), float("-inf")])
    arrays["post/x"] = x.clone()
    arrays["post/u8"] = du.postprocess(x.clone())                # [B,C,H,W] uint8
    save("image_pipeline", {}, arrays)


def distill_rich_case():
    """The RICH recipe in miniature (conf/{teacher,student,training}/rich.yaml): conditional 1-D flows (y_condition,
    3 condition features... y_classes=3), student K=8 L=2 / teacher K=8 L=3 -> EIGHT student hint layers against SIX teacher hint
    layers (pl_module.py:81-110; loss() zips them and divides by len(student_kd_indices), :262-282), loss weights
    0.85 / 0.075 (mse) / 0.075 (l1), per-sample `weights` multiplied into the loss (:309-310), y passed to student,
    teacher, prior and both reverse passes (:216-243).  Two steps, clip 30 + Adam."""
    tab = dict(is_1d=True, image_shape=[5], y_classes=3, y_condition=True, learn_top=False)
    s_cfg = base_cfg(K=8, L=2, hidden_channels=32, **tab)
    t_cfg = base_cfg(K=8, L=3, hidden_channels=64, **tab)
    student, s_sd = build(s_cfg, 151)
    teacher, t_sd = build(t_cfg, 153)
    teacher.eval()
    s_idx = [i for i in range(16) if (i + 1) % 2 == 0 or i + 1 == 16]
    t_idx = [i for i in range(24) if (i + 1) % 4 == 0 or i + 1 == 24]
    assert len(s_idx) == 8 and len(t_idx) == 6
    opt = torch.optim.Adam(student.parameters(), lr=5e-4, weight_decay=0)
    # parameters are rebuilt by the tests from (state_dict shapes, seed)
    arrays = {"s_seed": np.array(151), "t_seed": np.array(153)}
    arrays["s_idx"], arrays["t_idx"] = np.array(s_idx), np.array(t_idx)
    l1 = torch.nn.L1Loss(reduction="none")
    for step in range(2):
        g = torch.Generator().manual_seed(160 + step)
        B = 16
        x = torch.randn(B, 5, generator=g)
        y = torch.randn(B, 3, generator=g)          # RICH conditions on 3 real-valued features, not on a one-hot
        wts = torch.rand(B, generator=g) + 0.5
        arrays[f"step{step}/x"], arrays[f"step{step}/y"], arrays[f"step{step}/weights"] = x.clone(), y, wts
        opt.zero_grad()
        s_z, s_nll, _ = student(x, y)
        with torch.no_grad():
            t_z, _, _ = teacher(x, y)
        kd_val = None
        for si, ti in zip(s_idx, t_idx):
            d = torch.nn.functional.mse_loss(s_z[si], t_z[ti], reduction="none")
            d = d.mean(dim=list(range(d.dim()))[1:])
            kd_val = d if kd_val is None else kd_val + d
        kd_val = kd_val / len(s_idx)
        mean, logs = student.prior(x, y_onehot=y)
        torch.manual_seed(170 + step)
        latent = layers.gaussian_sample(mean, logs, 1)
        arrays[f"step{step}/latent"] = latent.clone()
        student_x = student(z=latent, temperature=0.7, reverse=True, y_onehot=y)[-1]
        with torch.no_grad():
            teacher_x = teacher(z=latent, temperature=0.7, reverse=True, y_onehot=y)[-1]
        perc = l1(student_x, teacher_x)
        perc = perc.mean(dim=list(range(perc.dim()))[1:])
        perc[(torch.isnan(perc)).nonzero(as_tuple=True)[0]] = 0
        res = 0.85 * s_nll + 0.075 * kd_val + 0.075 * perc
        res = res * wts
        res = res.mean()
        res.backward()
        gn = torch.nn.utils.clip_grad_norm_(student.parameters(), 30.0)
        for k, p in student.named_parameters():
            arrays[f"step{step}/grad/{k}"] = p.grad.clone() if p.grad is not None else torch.zeros_like(p)
        opt.step()
        arrays[f"step{step}/nll"], arrays[f"step{step}/kd"] = s_nll.mean(), kd_val.mean()
        arrays[f"step{step}/perceptual"], arrays[f"step{step}/loss"] = perc.mean(), res
        arrays[f"step{step}/grad_norm"] = gn
        for k, v in student.state_dict().items():
            arrays[f"step{step}/s_sd_after/{k}"] = v.clone()
    save("distill_rich5", {"student": s_cfg, "teacher": t_cfg}, arrays)


def round2_cases():
    primitive_case()     # now seeded end to end
    cifar = dict(image_shape=[32, 32, 3], hidden_channels=512, L=3)
    full_grads_case("cifar_k8_full_grads", base_cfg(K=8, **cifar), 4, 201)
    distill_full_case()
    inherit_case()
    image_pipeline_case()
    distill_rich_case()
    full_grads_case("celeba_l4_k32_full_grads",
                    base_cfg(image_shape=[64, 64, 3], hidden_channels=512, K=32, L=4), 2, 203)
    # CelebA student geometry of the reference (conf/student/celeba.yaml: K=16, hidden 256, L=3) at 64x64
    full_grads_case("celeba_student_k16_h256_grads",
                    base_cfg(image_shape=[64, 64, 3], hidden_channels=256, K=16, L=3), 2, 205)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if "--round2" in sys.argv:   # added in round 2: leaves the earlier fixtures (except primitives) untouched
        round2_cases()
        sys.exit(0)
    if "--only-reverse-grad-2d" in sys.argv:   # round 2, later: the image flows' reverse pass with gradients
        reverse_grad_2d_cases()
        sys.exit(0)
    if "--only-reverse-grad" in sys.argv:   # added after the other fixtures were committed: leaves them untouched
        reverse_grad_cases()
        distill_tab_case()
        sys.exit(0)
    primitive_case()
    model_case("glow8", base_cfg(), 2, 101)
    model_case("glow32", base_cfg(image_shape=[32, 32, 3], hidden_channels=32, L=3), 2, 102)
    model_case("glow8_additive", base_cfg(flow_coupling="additive"), 2, 103)
    model_case("glow8_nolu", base_cfg(LU_decomposed=False), 2, 104)
    model_case("glow8_shuffle", base_cfg(flow_permutation="shuffle"), 2, 105)
    model_case("glow8_reverse", base_cfg(flow_permutation="reverse"), 2, 106)
    model_case("glow8_learntop", base_cfg(learn_top=True), 2, 107)
    model_case("glow16_h24", base_cfg(image_shape=[16, 16, 3], hidden_channels=24, K=3, L=2), 3, 108)
    tab = dict(L=1, is_1d=True, y_classes=0)
    model_case("tab6", base_cfg(image_shape=[6], K=2, **tab), 8, 111)
    model_case("tab63", base_cfg(image_shape=[63], K=2, **tab), 8, 112)
    model_case("tab6_k5_h32", base_cfg(image_shape=[6], K=5, hidden_channels=32, **tab), 16, 113)
    model_case("tab5_cond", base_cfg(image_shape=[5], K=2, L=1, is_1d=True, y_classes=3,
                                     y_condition=True, learn_top=True), 8, 114)
    reverse_grad_cases()
    distill_case()
    distill_tab_case()
    cifar = dict(image_shape=[32, 32, 3], hidden_channels=512, L=3)
    checksum_case("cifar_k8_full", base_cfg(K=8, **cifar), 4, 201)
    checksum_case("cifar_k32_full", base_cfg(K=32, **cifar), 2, 202)
    checksum_case("celeba_l4_k32_full",
                  base_cfg(image_shape=[64, 64, 3], hidden_channels=512, K=32, L=4), 1, 203)
