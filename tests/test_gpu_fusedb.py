"""Fused FlowStep backward (csrc/nfk_fusedb.hip: dgrad3 -> ReLU/ActNorm bwd -> dgrad2 -> ... -> dgrad1 in ONE launch, then
the step's three weight-gradient GEMMs in ONE launch; NFK_BWD_FUSED, the default in bf16 mode) against the
per-conv launches it replaces (mode 2: one dgrad + wgrad launch per conv).

Both paths use the same bf16 roundings of the operands (weights, h1 / h2, g_a2 / g_a1 rounded once to bf16) and fp32
accumulation, but sum in different orders (k-steps of 16 over all taps at once here, 64-deep chunks per tap there), so a
few bf16 roundings of g_a2 / g_a1 flip: gradients agree to ~1e-3 of their largest entry, direction to 1e-5.  The pin to the
REFERENCE's gradients is tests/test_gpu_fullsize.py (which runs the default = fused path)."""
import pytest
import torch

from golden_util import Golden, rel_err
from param_util import randomize_state_dict, synthetic_images
from test_gpu_bf16 import _random_parameters, build
from test_oracle_golden import full_shapes

pytestmark = pytest.mark.gpu


def _grads(mode, cfg, sd, x, noise, seed=None, chain=True):
    from nf_distillation_amd import capi
    lib = capi.lib()
    try:
        model = build(cfg, sd, seed)
        model.flow.engine().chain_backward = chain
        model.flow.engine().bwd_mode = mode + 1   # nfk_step2d_grads.bwd_mode: 3 = NFK_BWD_DUAL, 4 = NFK_BWD_FUSED
        xin = x.clone().cuda().requires_grad_(True)
        outs, bpd, _ = model(xin, None, noise=noise.cuda())
        # a loss that also feeds gradients into intermediate layer outputs (the hint-MSE branch of the distillation loss)
        loss = bpd.mean() + 0.05 * sum(o.square().mean() for o in outs[::7])
        loss.backward()
        torch.cuda.synchronize()
        eng = model.flow.engine()
        steps = [r for r in eng.state["recs"] if r.kind == "step"]
        ok = [int(lib.nfk_step2d_fused_bwd_ok(__import__("ctypes").byref(r.cs))) for r in steps]
        grads = {k: p.grad.cpu() for k, p in model.named_parameters()}
        grads["__x__"] = xin.grad.cpu()
        return grads, ok
    finally:
        pass


def _compare(a, b, what):
    num = da = db = 0.0
    for k in a:
        u, v = a[k].double().flatten(), b[k].double().flatten()
        num, da, db = num + float(u @ v), da + float(u @ u), db + float(v @ v)
        if float(b[k].abs().max()) > 0:
            assert rel_err(a[k], b[k]) < 1e-2, (what, k, rel_err(a[k], b[k]))
    assert num / (da ** 0.5 * db ** 0.5) > 0.99999, what


def test_fused_backward_full_size_cifar():
    g = Golden("cifar_k8_full")
    cfg = g.cfg
    sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
    h, w, c = cfg["image_shape"]
    for B in (4, 5):   # 5: ragged tiles at level 3 (four 4x4 images per workgroup)
        x, noise = synthetic_images(B, (c, h, w), seed=int(g.t("seed")) + 9 + B)
        fused, ok = _grads(3, cfg, sd, x, noise)
        assert all(ok) and len(ok) == 24, ok           # every FlowStep of the model took the fused backward
        dual, _ = _grads(2, cfg, sd, x, noise)         # NFK_BWD_DUAL: one dgrad + wgrad launch per conv
        _compare(fused, dual, ("cifar", B))
        unchained, _ = _grads(3, cfg, sd, x, noise, chain=False)   # per-step calls (own coupling backward launch)
        _compare(unchained, dual, ("cifar unchained", B))


@pytest.mark.parametrize("variant", ["additive_B3", "shuffle_B6", "dense_invconv_B2"])
def test_fused_backward_variants(variant):
    from test_gpu_bf16 import VARIANTS
    v = dict(VARIANTS[variant])
    B = v.pop("B")
    cfg = {"image_shape": [32, 32, 3], "hidden_channels": 512, "K": 2, "L": 3, "actnorm_scale": 1.0, "y_classes": 10,
           "learn_top": False, "y_condition": False, "is_1d": False, **v}
    sd = _random_parameters(cfg, 11 + B)
    x, noise = synthetic_images(B, (3, 32, 32), seed=100 + B)
    fused, ok = _grads(3, cfg, sd, x, noise, seed=77)
    assert any(ok)
    dual, _ = _grads(2, cfg, sd, x, noise, seed=77)
    _compare(fused, dual, variant)


def test_fused_backward_celeba_student_falls_back():
    """hidden 256 (the reference's CelebA student): the fused kernels are instantiated for hidden 512 only; the step must
    run through the per-conv launches, not fail."""
    g = Golden("celeba_student_k16_h256_grads")
    cfg = dict(g.cfg, K=2)
    sd = _random_parameters(cfg, 3)
    x, noise = synthetic_images(1, (3, 64, 64), seed=5)
    fused, ok = _grads(3, cfg, sd, x, noise)
    assert not any(ok)
    dual, _ = _grads(2, cfg, sd, x, noise)
    for k in fused:
        assert torch.equal(fused[k], dual[k]), k


def test_fused_backward_weight_gradients_on_the_auxiliary_stream_bit_identical():
    """engine.defer_fused_wgrad: per step, the weight-gradient launch, its slab reductions and the ActNorm gradients leave
    the dgrad chain's stream for a chain of their own on the auxiliary stream (one fork per step, one join per backward;
    nfk_step2d_bwd_chain with ga / wslab).  Same kernels, same summation orders: every gradient bit for bit.  (Kept off by
    default: inside a replayed hipGraph the 24 extra cross-stream edges cost more than the concurrency returns.)"""
    from nf_distillation_amd import capi
    g = Golden("cifar_k8_full")
    cfg = g.cfg
    sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(4, (c, h, w), seed=int(g.t("seed")) + 21)
    res = []
    side = torch.cuda.Stream()
    aux = capi.lib().nfk_aux_create(side.cuda_stream)
    try:
        for defer in (False, True):
            model = build(cfg, sd)
            eng = model.flow.engine()
            eng.aux, eng.defer_fused_wgrad = (aux if defer else None), defer
            outs, bpd, _ = model(x.clone().cuda(), None, noise=noise.cuda())
            (bpd.mean() + 0.05 * sum(o.square().mean() for o in outs[::7])).backward()
            torch.cuda.synchronize()
            res.append({k: p.grad.clone() for k, p in model.named_parameters()})
    finally:
        capi.lib().nfk_aux_destroy(aux)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
