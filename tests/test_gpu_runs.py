"""A whole level's FlowSteps in ONE launch (nfk_run2d_fwd_fused, csrc/nfk_fused2.hip MULTI) against one launch per step.

Where a pixel tile is a whole image (CIFAR levels 2 and 3) the run kernel keeps z in LDS from step to step; every step's
arithmetic is the single-step kernel's, so everything -- layer outputs, bits/dim, what a training forward saves, and
therefore the gradients -- must be BIT-IDENTICAL to the per-step launches without hidden split.  The pin to the reference
is the golden bits/dim of the same full-size models (cifar_k32_full / cifar_k8_full: < 0.01 bpd in bf16 mode).
Also here: the KD-aware encode (`keep=`, SURVEY.md 8f-1 / pl_module.py:262-282)."""
import pytest
import torch

from golden_util import Golden
from param_util import randomize_state_dict, synthetic_images
from test_gpu_bf16 import build
from test_oracle_golden import full_shapes

pytestmark = pytest.mark.gpu


def _model_and_batch(name, batch=None):
    g = Golden(name)
    cfg = g.cfg
    sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
    h, w, c = cfg["image_shape"]
    B = int(g.t("batch")) if batch is None else batch
    x, noise = synthetic_images(B, (c, h, w), seed=int(g.t("seed")) + 5)
    return g, cfg, sd, x, noise


def _engine(model, runs, ways=0):
    eng = model.flow.engine()
    eng.fused_runs = runs
    eng.fused_split_inference = eng.fused_split_training = ways
    return eng


@pytest.mark.parametrize("batch", [None, 6])
def test_teacher_levels_as_runs_bit_identical(batch):
    """K = 32 teacher, inference: 3 x 32 launches -> 32 + 1 + 1.  batch 6: level 3 packs 4 images per tile, the last tile
    is half empty."""
    g, cfg, sd, x, noise = _model_and_batch("cifar_k32_full", batch)
    res = []
    for runs in (True, False):
        model = build(cfg, sd)
        eng = _engine(model, runs)
        with torch.no_grad():
            outs, bpd, _ = model(x.clone().cuda(), None, noise=noise.cuda())
        torch.cuda.synchronize()
        res.append((outs, bpd))
        if runs:
            assert sorted(v["n"] for v in eng.state["runs"].values()) == [32, 32]
    (o1, b1), (o0, b0) = res
    assert torch.equal(b1, b0)
    for i, (u, v) in enumerate(zip(o1, o0)):
        assert torch.equal(u, v), i
    if batch is None:
        assert float((b1.cpu() - g.t("bpd")).abs().max()) < 0.01   # ... and both sit on the reference's bits/dim


def test_training_forward_as_runs_bit_identical_gradients():
    """K = 8 student, forward + backward: the run kernel leaves z / h1 / h2 / ReLU masks / (shift, scale) for every step in
    strided arenas; the backward that reads them must give the very same gradients."""
    g, cfg, sd, x, noise = _model_and_batch("cifar_k8_full")
    res = []
    for runs in (True, False):
        model = build(cfg, sd)
        _engine(model, runs)
        xin = x.clone().cuda().requires_grad_(True)
        outs, bpd, _ = model(xin, None, noise=noise.cuda())
        loss = bpd.mean() + 0.05 * sum(o.square().mean() for o in outs[::7])   # gradients also enter at hint layers
        loss.backward()
        torch.cuda.synchronize()
        res.append((bpd.detach(), [o.detach() for o in outs], {k: p.grad.clone() for k, p in model.named_parameters()},
                    xin.grad.clone()))
    (b1, o1, g1, gx1), (b0, o0, g0, gx0) = res
    assert torch.equal(b1, b0)
    for i, (u, v) in enumerate(zip(o1, o0)):
        assert torch.equal(u, v), i
    assert torch.equal(gx1, gx0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
    assert float((b1.cpu() - g.t("bpd")).abs().max()) < 0.01


@pytest.mark.parametrize("runs", [True, False])
def test_kd_aware_encode_keeps_only_the_hint_layers(runs):
    """teacher(x, keep=[0, 34, 68, 100]) (the CIFAR hint layers, pl_module.py:81-110): same tensors at those indices, same
    bits/dim, None everywhere else."""
    g, cfg, sd, x, noise = _model_and_batch("cifar_k32_full")
    model = build(cfg, sd)
    _engine(model, runs, ways=4)
    keep = [0, 34, 68, 100]
    with torch.no_grad():
        full, bpd_full, _ = model(x.clone().cuda(), None, noise=noise.cuda())
        part, bpd_part, _ = model(x.clone().cuda(), None, noise=noise.cuda(), keep=keep)
        hint, bpd_hint, _ = model(x.clone().cuda(), None, noise=noise.cuda(), keep=keep, hints_only=True)
    assert len(part) == len(full) == len(hint) == 101
    assert torch.equal(bpd_full, bpd_part)
    assert bpd_hint is None   # hints only: nothing of the likelihood is computed (no Split2d prior conv, no log-det sums)
    for i in range(101):
        if i in keep:
            assert torch.equal(part[i], full[i]), i
            assert torch.equal(hint[i], full[i]), i
        else:
            assert part[i] is None and hint[i] is None, i
    # a pass that records gradients ignores `keep`: autograd may need any of the outputs
    student = build(cfg, sd)
    outs, _, _ = student(x.clone().cuda(), None, noise=noise.cuda(), keep=[-1])
    assert all(o is not None for o in outs)
