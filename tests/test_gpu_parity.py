"""Parity of the HIP path (through the C ABI) against the golden vectors captured from the
reference and against the CPU oracle, on a real MI355X.

Tolerances (fp32 path, north-star: log_prob and inverse samples within 1e-5 relative):
  * bits/dim (= log_prob up to a constant factor) : 1e-5 relative
  * layer outputs / inverse samples               : 1e-5 relative to the tensor's max magnitude
    (2e-5 for the deepest stacks, where the reference's own fp32-vs-fp64 drift is ~1e-6)
  * parameter gradients                            : 1e-3 relative to max (fp32 sums over
    B*H*W pixels in different orders; the oracle itself differs from the reference by ~1e-4)
"""
import pytest
import torch

from golden_util import Golden, rel_err
from oracle import glow_oracle as O
from param_util import randomize_state_dict, synthetic_images
from test_oracle_golden import full_shapes

pytestmark = pytest.mark.gpu

CASES_2D = ["glow8", "glow32", "glow8_additive", "glow8_nolu", "glow8_shuffle", "glow8_reverse",
            "glow8_learntop", "glow16_h24"]


def build(cfg, sd, g=None):
    import nf_distillation_amd as nfa
    model = nfa.create_glow_model(dict(cfg))
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    if g is not None:
        for i, idx in g.group("perm").items():
            layer = model.flow.layers[int(i)]
            (layer.shuffle if hasattr(layer, "shuffle") else layer.reverse).indices = idx.clone()
    return model.cuda()


@pytest.mark.parametrize("name", CASES_2D)
def test_forward_matches_reference(name):
    g = Golden(name)
    model = build(g.cfg, g.group("sd"), g)
    x = g.t("x").cuda()
    with torch.no_grad():
        outs, bpd, _ = model(x, None, noise=g.t("noise").cuda())
    torch.cuda.synchronize()
    # in-place dequantisation like models/utils.py:38
    assert torch.equal(x.cpu(), g.t("x") + g.t("noise"))
    ref = g.seq("out")
    assert len(outs) == len(ref)
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert tuple(o.shape) == tuple(r.shape), i
        assert rel_err(o.cpu(), r) < 2e-5, (i, rel_err(o.cpu(), r))
    assert rel_err(bpd.cpu(), g.t("bpd")) < 1e-5


@pytest.mark.parametrize("name", CASES_2D)
def test_backward_matches_reference(name):
    g = Golden(name)
    model = build(g.cfg, g.group("sd"), g)
    outs, bpd, _ = model(g.t("x").cuda(), None, noise=g.t("noise").cuda())
    bpd.mean().backward()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        ref = g.t(f"grad/{k}")
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(got.abs().max()) < 1e-6, k
        else:
            assert rel_err(got, ref) < 1e-3, (k, rel_err(got, ref))


@pytest.mark.parametrize("name", CASES_2D)
def test_reverse_matches_reference(name):
    g = Golden(name)
    model = build(g.cfg, g.group("sd"), g)
    z = g.seq("out")[-1].cuda()
    with torch.no_grad():
        rev = model(z=z, temperature=0.0, reverse=True)
    for i, (o, r) in enumerate(zip(rev, g.seq("rev"))):
        assert tuple(o.shape) == tuple(r.shape)
        assert rel_err(o.cpu(), r) < 5e-5, (i, rel_err(o.cpu(), r))


def test_round_trip_with_explicit_eps():
    """encode -> decode with the dropped halves re-injected as eps reproduces the input:
    a size-independent property (models/flows.py round trip, BASELINE.md: 1.4e-6 on CPU)."""
    g = Golden("glow32")
    cfg = g.cfg
    model = build(cfg, g.group("sd"))
    x = (g.t("x") + g.t("noise")).cuda()
    sd = g.group("sd")
    with torch.no_grad():
        outs, _, _ = model(x.clone(), None, noise=torch.zeros_like(x))
        # eps that reproduces the dropped half: (z2 - mean) / exp(logs), from the oracle on CPU
        eps = {}
        cpu_outs, _ = O.encode(sd, cfg, x.cpu(), O.dequant_logdet(x.cpu()))
        sched = O.layer_schedule(cfg)
        for i, (kind, _) in enumerate(sched):
            if kind == "split":
                xin = cpu_outs[i - 1]
                c = xin.shape[1]
                h = O.conv_zeros(sd, f"flow.layers.{i}.conv.", xin[:, : c // 2])
                eps[i] = ((xin[:, c // 2:] - h[:, 0::2]) / torch.exp(h[:, 1::2])).cuda()
        rev = model(z=outs[-1], temperature=1.0, reverse=True, eps=eps)
    assert rel_err(rev[-1].cpu(), x.cpu()) < 1e-5


@pytest.mark.parametrize("name", ["cifar_k8_full", "cifar_k32_full"])
def test_full_size_bpd_and_checksums(name):
    """BASELINE.json full sizes: Glow CIFAR-10 L=3 K=8 / K=32, hidden 512."""
    g = Golden(name)
    cfg = g.cfg
    sd = randomize_state_dict(full_shapes(cfg), int(g.t("seed")))
    model = build(cfg, sd)
    h, w, c = cfg["image_shape"]
    x, noise = synthetic_images(int(g.t("batch")), (c, h, w), seed=int(g.t("seed")) + 5)
    with torch.no_grad():
        outs, bpd, _ = model(x.cuda(), None, noise=noise.cuda())
    assert rel_err(bpd.cpu(), g.t("bpd")) < 1e-5, rel_err(bpd.cpu(), g.t("bpd"))
    got = torch.stack([o.double().abs().mean() for o in outs]).cpu()
    assert rel_err(got, g.t("out_absmean")) < 1e-5


def test_logdet_is_deterministic():
    g = Golden("glow32")
    model = build(g.cfg, g.group("sd"))
    res = []
    for _ in range(3):
        with torch.no_grad():
            _, bpd, _ = model(g.t("x").cuda(), None, noise=g.t("noise").cuda())
        res.append(bpd.cpu())
    assert torch.equal(res[0], res[1]) and torch.equal(res[1], res[2])


CASES_1D = ["tab6", "tab63", "tab6_k5_h32", "tab5_cond"]


@pytest.fixture
def tab_layout(request):
    """resident: weights in LDS for the life of a workgroup, contractions on exact-f32 MFMA (the default where it fits);
    staged: the per-tile staging on the vector ALUs (what steps wider than 64 features or hidden units fall back to)."""
    from nf_distillation_amd import capi
    capi.lib().nfk_tune(capi.TUNE_TAB_RESIDENT, 1 if request.param == "resident" else 0)
    yield request.param
    capi.lib().nfk_tune(capi.TUNE_TAB_RESIDENT, 1)


@pytest.mark.parametrize("tab_layout", ["resident", "staged"], indirect=True)
@pytest.mark.parametrize("name", CASES_1D)
def test_tabular_forward_backward_reverse(name, tab_layout):
    """1-D Glow (BASELINE.json configs 1 and 5 map onto this path: SURVEY.md §0 F1)."""
    g = Golden(name)
    model = build(g.cfg, g.group("sd"))
    x = g.t("x").cuda()
    y = g.t("y").cuda() if g.has("y") else None
    outs, nll, y_logits = model(x, y)
    for i, (o, r) in enumerate(zip(outs, g.seq("out"))):
        assert rel_err(o.cpu(), r) < 2e-5, (i, rel_err(o.cpu(), r))
    assert rel_err(nll.cpu(), g.t("bpd")) < 1e-5
    loss = nll.mean()
    if y_logits is not None:
        assert rel_err(y_logits.cpu(), g.t("y_logits")) < 2e-5
        loss = loss + 0.1 * (y_logits ** 2).sum()
    loss.backward()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        ref = g.t(f"grad/{k}")
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(got.abs().max()) < 1e-6, k
        else:
            assert rel_err(got, ref) < 1e-3, (k, rel_err(got, ref))
    with torch.no_grad():
        rev = model(z=outs[-1].detach(), y_onehot=y, temperature=0.0, reverse=True)
    for i, (o, r) in enumerate(zip(rev, g.seq("rev"))):
        assert rel_err(o.cpu(), r) < 1e-4, (i, rel_err(o.cpu(), r))
    # exact inverse: the last reverse output reproduces the input rows
    assert rel_err(rev[-1].cpu(), g.t("x")) < 1e-4


def test_tabular_large_batch_round_trip():
    """POWER-shaped batch of 65 536 rows (conf/training/tabular.yaml:7): encode -> decode."""
    g = Golden("tab6_k5_h32")
    model = build(g.cfg, g.group("sd"))
    torch.manual_seed(0)
    x = torch.randn(65536, 6, device="cuda")
    with torch.no_grad():
        outs, nll, _ = model(x.clone(), None)
        rev = model(z=outs[-1], temperature=1.0, reverse=True)
    assert rel_err(rev[-1].cpu(), x.cpu()) < 1e-4
    assert torch.isfinite(nll).all()


def test_tabular_layouts_agree_at_65536_rows():
    """The two layouts of the 1-D kernels (resident + MFMA, tall tiles / staged + vector ALUs) on the large batch the fixtures
    do not reach: d = 63 K = 3, 65 536 rows + a ragged tail -- bits/dim 1e-5, every gradient 1e-4 of its largest entry."""
    from nf_distillation_amd import capi
    import nf_distillation_amd as nfa
    cfg = dict(Golden("tab63").cfg)
    cfg["K"] = 3
    torch.manual_seed(5)
    x = torch.randn(65536 + 37, 63, device="cuda")
    res = {}
    try:
        for mode in (1, 0):
            capi.lib().nfk_tune(capi.TUNE_TAB_RESIDENT, mode)
            torch.manual_seed(9)
            m = nfa.create_glow_model(dict(cfg)).cuda()
            for p in m.parameters():
                if p.dim() >= 1 and p.numel() > 1 and p.requires_grad:
                    p.data.add_(0.05 * torch.randn_like(p))
            _, nll, _ = m(x.clone(), None)
            nll.mean().backward()
            torch.cuda.synchronize()
            res[mode] = (nll.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    finally:
        capi.lib().nfk_tune(capi.TUNE_TAB_RESIDENT, 1)
    assert rel_err(res[1][0].cpu(), res[0][0].cpu()) < 1e-5
    for k, gref in res[0][1].items():
        if float(gref.abs().max()) > 0:
            assert rel_err(res[1][1][k].cpu(), gref.cpu()) < 1e-4, (k, rel_err(res[1][1][k].cpu(), gref.cpu()))


REVGRAD_CASES = ["tab6_revgrad", "tab63_revgrad", "tab6_k5_h32_revgrad", "tab6_additive_nolu_revgrad",
                 "tab5_cond_revgrad"]


@pytest.mark.parametrize("name", REVGRAD_CASES)
def test_tabular_reverse_pass_gradients(name):
    """Reverse (sampling) pass WITH gradients through the HIP kernels (nfk_tab_inv_bwd + nfk_prep_inv_bwd): the
    perceptual branch of the tabular distillation loss, pl_module.py:229-243, 283-299.  Samples, the L1 loss and the
    gradients of every parameter and of the latent against the reference's."""
    g = Golden(name)
    model = build(g.cfg, g.group("sd"))
    z = g.t("z").cuda().requires_grad_(True)
    y = g.t("y").cuda() if g.has("y") else None
    outs = model(z=z, y_onehot=y, temperature=0.7, reverse=True)
    for i, (o, r) in enumerate(zip(outs, g.seq("rev"))):
        assert rel_err(o.detach().cpu(), r) < 1e-4, (i, rel_err(o.detach().cpu(), r))
    loss = torch.nn.L1Loss(reduction="none")(outs[-1], g.t("target").cuda()).mean(dim=1).mean()
    assert rel_err(loss.detach().cpu(), g.t("loss")) < 1e-5
    loss.backward()
    torch.cuda.synchronize()
    assert rel_err(z.grad.cpu(), g.t("grad_z")) < 1e-3, rel_err(z.grad.cpu(), g.t("grad_z"))
    for k, p in model.named_parameters():
        ref = g.t(f"grad/{k}")
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)
        if float(ref.abs().max()) == 0.0:
            assert float(got.abs().max()) < 1e-6, k
        else:
            assert rel_err(got, ref) < 1e-3, (k, rel_err(got, ref))


REVGRAD_2D_CASES = ["glow8_revgrad", "glow16_additive_nolu_revgrad", "glow8_shuffle_revgrad"]


@pytest.mark.parametrize("name", REVGRAD_2D_CASES)
def test_image_reverse_pass_gradients(name):
    """The image flows' reverse pass WITH gradients through the HIP kernels (nfk_step2d_inv + nfk_step2d_inv_bwd +
    nfk_prep_inv_bwd; pl_module.py:229-243 with an image student): samples, the L1 loss, and the gradients of every
    parameter and of the latent against the reference's (Split2d's torch.normal draw reproduced from its stored N(0,1)
    variates; it carries no gradient upstream, so the prior convs' gradients must be exactly zero)."""
    g = Golden(name)
    model = build(g.cfg, g.group("sd"), g)
    z = g.t("z").cuda().requires_grad_(True)
    eps = {int(k): v.cuda() for k, v in g.group("eps").items()}
    outs = model(z=z, y_onehot=None, temperature=0.7, reverse=True, eps=eps)
    for i, (o, r) in enumerate(zip(outs, g.seq("rev"))):
        assert rel_err(o.detach().cpu(), r) < 1e-4, (i, rel_err(o.detach().cpu(), r))
    per = torch.nn.L1Loss(reduction="none")(outs[-1], g.t("target").cuda())
    loss = per.mean(dim=(1, 2, 3)).mean()
    assert rel_err(loss.detach().cpu(), g.t("loss")) < 1e-5
    loss.backward()
    torch.cuda.synchronize()
    assert rel_err(z.grad.cpu(), g.t("grad_z")) < 1e-3, rel_err(z.grad.cpu(), g.t("grad_z"))
    for k, p in model.named_parameters():
        ref = g.t(f"grad/{k}")
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)
        if float(ref.abs().max()) == 0.0:
            assert float(got.abs().max()) < 1e-6, k
        else:
            assert rel_err(got, ref) < 1e-3, (k, rel_err(got, ref))
