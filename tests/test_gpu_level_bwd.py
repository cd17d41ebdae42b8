"""Backward of a whole level (nfk_level2d_bwd, csrc/nfk_fusedb.hip level mode + level_wgrad_kernel / level_tail_kernel)
against the per-step calls it replaces (nfk_step2d_bwd_chain: fused dgrad chain, merged weight-gradient / channel-mix
launch), which tests/test_gpu_fusedb.py and tests/test_gpu_fullsize.py pin to the reference.

Both forms run the same dgrad chain (same MFMA order: g_a2 / g_a1 / the im2col rows are the same bits) and the same fp32
channel-mix / coupling arithmetic per element; what differs is the ORDER of fp32 sums over pixels -- split-K ranges of the
weight-gradient GEMMs (a quarter of the pixels per slab instead of a sixteenth), partial rows per dgrad workgroup instead
of per channel-mix workgroup.  Bars, relative to the tensor's largest entry (the per-step form against the reference: 1e-3):
  one launch per step : input gradient 1e-6, parameter gradients 2e-5;
  one launch per level: input gradient 2e-4 (observed 3.8e-5 at B=64, K=8: the instantiation that walks the steps in a loop
                        contracts a few fp32 multiply-adds of the coupling backward differently; one ulp in g_a3 flips a
                        bf16 rounding of the next dgrad's operand now and then, 24 steps deep), parameter gradients 2e-5 --
                        and two runs of it must agree bit for bit (a race between the steps of a run would not)."""
import pytest
import torch

from param_util import randomize_state_dict, shapes_of, synthetic_images

pytestmark = pytest.mark.gpu


def _cfg(K, **kw):
    cfg = dict(image_shape=[32, 32, 3], hidden_channels=512, K=K, L=3, actnorm_scale=1.0, flow_permutation="invconv",
               flow_coupling="affine", LU_decomposed=True, y_classes=10, learn_top=False, y_condition=False, is_1d=False)
    cfg.update(kw)
    return cfg


def _grads(cfg, B, seed, mode, hints):
    import nf_distillation_amd as nfa
    torch.manual_seed(1234)   # (the `shuffle` permutation is drawn at construction and is not part of the state_dict)
    m = nfa.create_glow_model(cfg)
    m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), 11))
    m = m.cuda().set_compute_dtype("bf16")
    eng = m.flow.engine()
    eng.level_backward = mode != "steps"
    eng.level_multi = mode == "level_multi"
    x, noise = synthetic_images(B, seed=seed)
    xin = x.cuda().requires_grad_(True)
    outs, bpd, _ = m(xin, None, noise=noise.cuda())
    loss = bpd.mean()
    if hints:   # gradients into layer outputs inside the levels too: a level is then walked in several calls
        loss = loss + 0.05 * sum(o.square().mean() for o in outs[::hints])
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.clone() for k, p in m.named_parameters()}
    g["__x__"] = xin.grad.clone()
    return g


def _compare(a, b, what, x_tol=1e-6):
    worst = (0.0, "")
    for k in b:
        scale = float(b[k].abs().max())
        if scale == 0.0:
            assert float(a[k].abs().max()) == 0.0, (what, k)
            continue
        err = float((a[k] - b[k]).abs().max()) / scale
        worst = max(worst, (err, k))
        assert err <= (x_tol if k == "__x__" else 2e-5), (what, k, err)
    print(what, "worst", worst)


@pytest.mark.parametrize("B,K,hints", [(4, 3, 0), (5, 2, 3), (64, 8, 0), (64, 3, 7)])
def test_level_backward_equals_per_step_backward(B, K, hints):
    cfg = _cfg(K)
    ref = _grads(cfg, B, 40 + B, "steps", hints)
    _compare(_grads(cfg, B, 40 + B, "level", hints), ref, ("level", B, K, hints))
    multi = _grads(cfg, B, 40 + B, "level_multi", hints)
    _compare(multi, ref, ("level_multi", B, K, hints), x_tol=2e-4)
    again = _grads(cfg, B, 40 + B, "level_multi", hints)
    for k in multi:
        assert torch.equal(multi[k], again[k]), ("run to run", k)


@pytest.mark.parametrize("variant", [dict(flow_coupling="additive"), dict(flow_permutation="shuffle"), dict(LU_decomposed=False)])
def test_level_backward_variants(variant):
    cfg = _cfg(2, **variant)
    ref = _grads(cfg, 6, 77, "steps", 5)
    _compare(_grads(cfg, 6, 77, "level", 5), ref, ("level", variant))
    _compare(_grads(cfg, 6, 77, "level_multi", 5), ref, ("level_multi", variant), x_tol=2e-4)


def test_level_backward_on_the_auxiliary_stream_is_bit_identical():
    import nf_distillation_amd as nfa
    from nf_distillation_amd import capi
    cfg = _cfg(3)
    x, noise = synthetic_images(8, seed=5)
    res = []
    side = torch.cuda.Stream()
    aux = capi.lib().nfk_aux_create(side.cuda_stream)
    try:
        for use_aux in (False, True):
            m = nfa.create_glow_model(cfg)
            m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), 11))
            m = m.cuda().set_compute_dtype("bf16")
            eng = m.flow.engine()
            eng.aux, eng.level_wgrad_aux = (aux if use_aux else None), use_aux   # (opt-in: engine.py)
            _, bpd, _ = m(x.clone().cuda(), None, noise=noise.cuda())
            bpd.mean().backward()
            torch.cuda.synchronize()
            res.append({k: p.grad.clone() for k, p in m.named_parameters()})
    finally:
        capi.lib().nfk_aux_destroy(aux)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
