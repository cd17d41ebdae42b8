"""Throughput mode (bf16 operands, fp32 accumulation in the coupling-net convolutions).

The bar for this mode is BASELINE.json's: student NLL bits/dim within 0.01 of the reference.
Everything else (ActNorm/invconv mixing, coupling tail, log-det sums, priors, optimiser) stays
fp32, so the deviation comes only from rounding the 512-wide hidden tensors and the conv weights
to bf16 (8 significant bits)."""
import pytest
import torch

from golden_util import Golden, rel_err
from param_util import randomize_state_dict, synthetic_images
from test_oracle_golden import full_shapes

pytestmark = pytest.mark.gpu


def build(cfg, sd, seed=None):
    import nf_distillation_amd as nfa
    if seed is not None:   # `shuffle` permutations are drawn at construction and are not part of the state_dict
        torch.manual_seed(seed)
    model = nfa.create_glow_model(dict(cfg))
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    return model.cuda().set_compute_dtype("bf16")


@pytest.mark.parametrize("name", ["cifar_k8_full", "cifar_k32_full"])
def test_full_size_bpd_within_0p01(name):
    g = Golden(name)
    cfg = g.cfg
    model = build(cfg, randomize_state_dict(full_shapes(cfg), int(g.t("seed"))))
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(int(g.t("batch")), (c, h, w), seed=int(g.t("seed")) + 5)
    with torch.no_grad():
        outs, bpd, _ = model(x.cuda(), None, noise=noise.cuda())
    err = float((bpd.cpu() - g.t("bpd")).abs().max())
    assert err < 0.01, err
    got = torch.stack([o.double().abs().mean() for o in outs]).cpu()
    assert rel_err(got, g.t("out_absmean")) < 2e-2


@pytest.mark.parametrize("name", ["glow32", "glow16_h24", "glow8_additive"])
def test_small_models_forward_backward_reverse(name):
    g = Golden(name)
    model = build(g.cfg, g.group("sd"))
    outs, bpd, _ = model(g.t("x").cuda(), None, noise=g.t("noise").cuda())
    assert float((bpd.cpu() - g.t("bpd")).abs().max()) < 0.01
    for o, r in zip(outs, g.seq("out")):
        assert rel_err(o.cpu(), r) < 3e-2
    bpd.mean().backward()
    torch.cuda.synchronize()
    num = den_a = den_b = 0.0
    for k, p in model.named_parameters():
        ref = g.t(f"grad/{k}").double().flatten()
        got = p.grad.cpu().double().flatten()
        num += float(ref @ got)
        den_a += float(ref @ ref)
        den_b += float(got @ got)
    cos = num / (den_a ** 0.5 * den_b ** 0.5)
    assert cos > 0.999, cos
    with torch.no_grad():
        rev = model(z=g.seq("out")[-1].cuda(), temperature=0.0, reverse=True)
    assert rel_err(rev[-1].cpu(), g.seq("rev")[-1]) < 3e-2


@pytest.mark.parametrize("name", ["glow8_revgrad", "glow16_additive_nolu_revgrad"])
def test_image_reverse_pass_gradients_bf16(name):
    """The reverse pass with gradients (nfk_step2d_inv_bwd) with bf16 GEMM operands: sample within 3e-2 of the reference's,
    global gradient cosine > 0.995 (measured 0.9986 on the hidden-16 toy: the inverse coupling divides by the scale, which
    amplifies the bf16 rounding of the coupling net a little more than the forward direction's 0.999)."""
    g = Golden(name)
    model = build(g.cfg, g.group("sd"))
    z = g.t("z").cuda().requires_grad_(True)
    eps = {int(k): v.cuda() for k, v in g.group("eps").items()}
    outs = model(z=z, y_onehot=None, temperature=0.7, reverse=True, eps=eps)
    assert rel_err(outs[-1].detach().cpu(), g.seq("rev")[-1]) < 3e-2
    (outs[-1] - g.t("target").cuda()).abs().mean(dim=(1, 2, 3)).mean().backward()
    torch.cuda.synchronize()
    num = den_a = den_b = 0.0
    for k, p in list(model.named_parameters()) + [("z", z)]:
        ref = (g.t("grad_z") if k == "z" else g.t(f"grad/{k}")).double().flatten()
        got = (p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)).double().flatten()
        num += float(ref @ got)
        den_a += float(ref @ ref)
        den_b += float(got @ got)
    cos = num / (den_a ** 0.5 * den_b ** 0.5)
    assert cos > 0.995, cos


def test_distillation_step_trains():
    """A few bf16 steps on the tiny distillation pair track the fp32 reference trace loosely and reduce the loss."""
    from nf_distillation_amd.training import DistillationStep
    g = Golden("distill8")
    import nf_distillation_amd as nfa
    student = nfa.create_glow_model(dict(g.cfg["student"]))
    student.load_state_dict(g.group("s_sd"))
    teacher = nfa.create_glow_model(dict(g.cfg["teacher"]))
    teacher.load_state_dict(g.group("t_sd"))
    student, teacher = student.cuda().set_compute_dtype("bf16"), teacher.cuda().set_compute_dtype("bf16")
    step = DistillationStep(student, teacher)
    out = step.training_step(g.t("step0/x").cuda(), g.t("step0/noise_s").cuda(), g.t("step0/noise_t").cuda())
    assert abs(float(out["nll"]) - float(g.t("step0/nll"))) < 0.01
    assert rel_err(out["result_loss"].cpu(), g.t("step0/loss")) < 5e-3
    first = float(out["result_loss"])
    x0 = g.t("step0/x")
    for _ in range(20):
        out = step.training_step(x0.clone().cuda(), torch.zeros_like(x0).cuda(), torch.zeros_like(x0).cuda())
    assert float(out["result_loss"]) < first


@pytest.mark.parametrize("ways", [4, 0, 2], ids=["split4", "no_hidden_split", "split2"])
@pytest.mark.parametrize("name", ["cifar_k8_full", "glow32"])
def test_fused_step_equals_unfused(name, ways):
    """The one-launch FlowStep forward (csrc/nfk_fused2.hip) against the five-kernel bf16 path: same bf16 roundings of
    h1/h2/weights, so outputs agree to fp32 summation-order noise; gradients (same backward) likewise.  `ways`: the hidden
    split the fused launch uses on small grids (CIFAR levels 2 / 3): 4-way (default policy), none, 2-way."""
    _fused_vs_unfused(name, ways)


def _fused_vs_unfused(name, ways=4):
    g = Golden(name)
    cfg = g.cfg
    if name.endswith("_full"):
        sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
        h, w, c = cfg["image_shape"]
        x, noise = synthetic_images(int(g.t("batch")), (c, h, w), seed=int(g.t("seed")) + 5)
    else:
        sd, x, noise = g.group("sd"), g.t("x"), g.t("noise")
    _compare_fused_unfused(cfg, sd, x, noise, ways=ways)


def _compare_fused_unfused(cfg, sd, x, noise, seed=None, ways=4):
    res = []
    for fused in (True, False):
        model = build(cfg, sd, seed).set_fused(fused)
        model.flow.engine().fused_split_training = ways
        outs, bpd, _ = model(x.clone().cuda(), None, noise=noise.cuda())
        bpd.mean().backward()
        torch.cuda.synchronize()
        res.append(([o.detach().cpu() for o in outs], bpd.detach().cpu(), {k: p.grad.cpu() for k, p in model.named_parameters()}))
    (o1, b1, g1), (o0, b0, g0) = res
    assert rel_err(b1, b0) < 2e-5, rel_err(b1, b0)
    for i, (a, b) in enumerate(zip(o1, o0)):
        assert rel_err(a, b) < 2e-3, (i, rel_err(a, b))
    # the two forwards round h1/h2 identically but sum in different orders, so a few ReLU masks / bf16 roundings
    # flip; gradients of a 29-layer chain then differ by a few % on individual tensors but point the same way
    num = da = db = 0.0
    for k in g0:
        a_, b_ = g1[k].double().flatten(), g0[k].double().flatten()
        num, da, db = num + float(a_ @ b_), da + float(a_ @ a_), db + float(b_ @ b_)
        if float(g0[k].abs().max()) > 0:
            assert rel_err(g1[k], g0[k]) < 0.1, (k, rel_err(g1[k], g0[k]))
    assert num / (da ** 0.5 * db ** 0.5) > 0.9995


def _random_parameters(cfg, seed):
    """state_dict of a model of this configuration with every parameter drawn at random (the reference initialises the
    last conv of each coupling net to zero, which would make the coupling trivial); buffers keep their initial values"""
    import math
    import nf_distillation_amd as nfa
    torch.manual_seed(seed)   # the invconv buffers (p, sign_s) and shuffle indices come from the constructor's draws
    model = nfa.create_glow_model(dict(cfg))
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            leaf = k.split(".")[-1]
            r = torch.randn(p.shape, generator=gen)
            if leaf in ("lower", "upper"):
                p.copy_(r * 0.3 / math.sqrt(p.shape[0]))
            elif k.endswith("invconv.weight"):
                q, _ = torch.linalg.qr(torch.randn(p.shape, generator=gen, dtype=torch.float64))
                p.copy_(q.float() + 0.05 * r)
            elif leaf == "weight":      # conv weights [out, in, kh, kw]
                p.copy_(r * 0.7 / math.sqrt(p.shape[1] * p.shape[2] * p.shape[3]))
            else:                       # biases, logs, log_s, prior
                p.copy_(r * 0.1)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


VARIANTS = {
    # hidden 512 = the fused kernel's shape; batches that leave ragged tiles at level 3 (4 images per workgroup)
    "affine_B5": dict(flow_coupling="affine", flow_permutation="invconv", LU_decomposed=True, B=5),
    "additive_B3": dict(flow_coupling="additive", flow_permutation="invconv", LU_decomposed=True, B=3),
    "shuffle_B6": dict(flow_coupling="affine", flow_permutation="shuffle", LU_decomposed=True, B=6),
    "reverse_dense_B1": dict(flow_coupling="affine", flow_permutation="reverse", LU_decomposed=False, B=1),
    "dense_invconv_B2": dict(flow_coupling="affine", flow_permutation="invconv", LU_decomposed=False, B=2),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_fused_step_variants(variant):
    """Coupling type, permutation type and ragged batches through the one-launch FlowStep (hidden 512, K=2, L=3, 32x32):
    against the five-kernel bf16 path (same roundings) and, for bits/dim, against this engine's fp32 parity mode (which
    the fp32 tests pin to the oracle at 1e-5): BASELINE.json's 0.01 bpd bar."""
    import nf_distillation_amd as nfa
    v = dict(VARIANTS[variant])
    B = v.pop("B")
    cfg = {"image_shape": [32, 32, 3], "hidden_channels": 512, "K": 2, "L": 3, "actnorm_scale": 1.0, "y_classes": 10,
           "learn_top": False, "y_condition": False, "is_1d": False, **v}
    sd = _random_parameters(cfg, 11 + B)
    x, noise = synthetic_images(B, (3, 32, 32), seed=100 + B)
    _compare_fused_unfused(cfg, sd, x, noise, seed=77)
    torch.manual_seed(77)
    m32 = nfa.create_glow_model(dict(cfg))
    m32.load_state_dict({k: t.clone() for k, t in sd.items()})
    with torch.no_grad():
        _, bpd32, _ = m32.cuda()(x.clone().cuda(), None, noise=noise.cuda())
        _, bpd16, _ = build(cfg, sd, 77)(x.clone().cuda(), None, noise=noise.cuda())
    assert float((bpd16 - bpd32).abs().max()) < 0.01, (bpd16, bpd32)


def test_fused_step_on_32_pixel_rows():
    """CelebA-64 geometry: level 1 is 32 x 32 (rows of 32 pixels): one image row (+ a halo row each side) per workgroup --
    32 output pixels, 96 computed -- through the one-launch forward, the fused backward per level (and the per-step calls)
    against the five-kernel bf16 path, as the CIFAR geometries above; bits/dim against fp32 parity mode (0.01)."""
    import nf_distillation_amd as nfa
    from nf_distillation_amd import capi
    import ctypes
    cfg = {"image_shape": [64, 64, 3], "hidden_channels": 512, "K": 2, "L": 2, "actnorm_scale": 1.0, "y_classes": 10,
           "learn_top": False, "y_condition": False, "is_1d": False, "flow_coupling": "affine", "flow_permutation": "invconv",
           "LU_decomposed": True}
    sd = _random_parameters(cfg, 21)
    x, noise = synthetic_images(3, (3, 64, 64), seed=64)
    _compare_fused_unfused(cfg, sd, x, noise, seed=77)
    model = build(cfg, sd, 77)
    outs, bpd16, _ = model(x.clone().cuda(), None, noise=noise.cuda())
    steps = [r for r in model.flow.engine().state["recs"] if r.kind == "step"]
    assert all(capi.lib().nfk_step2d_fused_tiles(ctypes.byref(r.cs)) > 0 for r in steps)          # level 1 included
    # backward: level 1 (32 x 32 x 12) on the fused kernels, one image row per tile; level 2 here (16 x 16 x 24: with four
    # rows per tile a 96-pixel h-region with 256-wide im2col rows, 164 KB of LDS) on tiles of two rows
    assert all(capi.lib().nfk_step2d_fused_bwd_ok(ctypes.byref(r.cs)) for r in steps)
    assert all(capi.lib().nfk_level_bwd_ok(ctypes.byref(r.cs)) == 1 for r in steps)
    assert capi.lib().nfk_step2d_fused_tiles(ctypes.byref(steps[0].cs)) == 32                       # one tile per image row
    # the level form of the backward against the per-step calls on this geometry
    grads = {}
    for lvl in (True, False):
        m = build(cfg, sd, 77)
        m.flow.engine().level_backward = lvl
        _, bpd, _ = m(x.clone().cuda(), None, noise=noise.cuda())
        bpd.mean().backward()
        torch.cuda.synchronize()
        grads[lvl] = {k: p.grad.clone() for k, p in m.named_parameters()}
    for k, v in grads[False].items():
        if float(v.abs().max()) > 0:
            assert rel_err(grads[True][k], v) < 2e-5, (k, rel_err(grads[True][k], v))
    torch.manual_seed(77)
    m32 = nfa.create_glow_model(dict(cfg))
    m32.load_state_dict({k: t.clone() for k, t in sd.items()})
    with torch.no_grad():
        _, bpd32, _ = m32.cuda()(x.clone().cuda(), None, noise=noise.cuda())
    assert float((bpd16.detach() - bpd32).abs().max()) < 0.01, (bpd16, bpd32)


def _grads_with(mode, halo, g, cfg, sd, x, noise, chain=False):
    """mode: 0 separate launches, 1 dual launches for small steps, 2 dual launches (nfk_step2d_grads.bwd_mode - 1)"""
    from nf_distillation_amd import capi
    lib = capi.lib()
    lib.nfk_tune(capi.TUNE_NARROW_HALO_DGRAD, 1 if halo else 0)   # conv1 dgrad through the halo-tile kernel or the split-K GEMM
    lib.nfk_tune(capi.TUNE_WGRAD_HALO, 1 if halo else 0)          # conv3 weight gradient through the halo-tile kernel or the generic one
    try:
        model = build(cfg, sd)
        model.flow.engine().chain_backward = chain
        model.flow.engine().bwd_mode = mode + 1
        outs, bpd, _ = model(x.clone().cuda(), None, noise=noise.cuda())
        bpd.mean().backward()
        torch.cuda.synchronize()
        return {k: p.grad.cpu() for k, p in model.named_parameters()}
    finally:
        lib.nfk_tune(capi.TUNE_NARROW_HALO_DGRAD, 1)
        lib.nfk_tune(capi.TUNE_WGRAD_HALO, 1)


def test_dual_dgrad_wgrad_launch_is_bit_identical():
    """bf16 backward: a conv's dgrad GEMM and weight-gradient GEMM run as ONE launch (NFK_BWD_DUAL) or as
    two (NFK_BWD_SEPARATE).  Same tiles, same split-K, same summation order: every gradient must agree bit for bit.  (With the
    halo-tile kernel for conv1's dgrad, which sums the taps in another order, the gradients agree to rounding: next test.)"""
    g = Golden("cifar_k8_full")
    cfg = g.cfg
    sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(int(g.t("batch")), (c, h, w), seed=int(g.t("seed")) + 9)
    res = [_grads_with(mode, False, g, cfg, sd, x, noise) for mode in (2, 0, 1)]
    for other in res[1:]:
        for k in res[0]:
            assert torch.equal(res[0][k], other[k]), k


@pytest.mark.parametrize("coupling", ["affine", "additive"])
def test_chained_backward_equals_per_step_calls(coupling):
    """nfk_step2d_bwd_chain folds the next step's coupling backward into the launch that ends the current step's.  The
    element-wise results are the same numbers; only the per-channel sums behind block.4's bias / logs gradients are
    reduced over pixel tiles instead of over samples.  Everything else must agree bit for bit."""
    g = Golden("cifar_k8_full")
    cfg = g.cfg
    if coupling == "affine":
        sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
        B = int(g.t("batch"))
    else:   # additive coupling (Cout = C/2), a batch that leaves ragged tiles
        cfg = dict(cfg, flow_coupling="additive", K=3)
        sd = _random_parameters(cfg, 5)
        B = 5
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(B, (c, h, w), seed=int(g.t("seed")) + 9)
    a = _grads_with(2, True, g, cfg, sd, x, noise, chain=True)
    b = _grads_with(2, True, g, cfg, sd, x, noise, chain=False)
    n_sum = 0
    for k in a:
        if k.endswith("block.4.conv.bias") or k.endswith("block.4.logs"):
            assert rel_err(a[k], b[k]) < 1e-5, (k, rel_err(a[k], b[k]))
            n_sum += 1
        else:
            assert torch.equal(a[k], b[k]), k
    assert n_sum > 0


def test_halo_tile_kernels_agree_with_generic_gemms():
    """conv1's dgrad and conv3's weight gradient through the halo-tile kernels (wide input tile staged once in LDS, 9 taps
    as shifted rows) against the implicit-GEMM kernels that fetch the input once per tap: same products, other
    summation order -- a few bf16 roundings of the downstream gradients flip, nothing more."""
    g = Golden("cifar_k8_full")
    cfg = g.cfg
    sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(int(g.t("batch")), (c, h, w), seed=int(g.t("seed")) + 9)
    a, b = _grads_with(2, True, g, cfg, sd, x, noise), _grads_with(2, False, g, cfg, sd, x, noise)
    num = da = db = 0.0
    for k in a:
        u, v = a[k].double().flatten(), b[k].double().flatten()
        num, da, db = num + float(u @ v), da + float(u @ u), db + float(v @ v)
        if float(b[k].abs().max()) > 0:
            assert rel_err(a[k], b[k]) < 2e-2, (k, rel_err(a[k], b[k]))
    assert num / (da ** 0.5 * db ** 0.5) > 0.99999
