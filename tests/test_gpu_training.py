"""Distillation training_step on the GPU against the golden two-step trace captured from the
reference models driven by torch.optim.Adam + clip_grad_norm_ (tests/golden/make_golden.py)."""
import pytest
import torch

from golden_util import Golden, rel_err

pytestmark = pytest.mark.gpu


def _build(cfg, sd):
    import nf_distillation_amd as nfa
    m = nfa.create_glow_model(dict(cfg))
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.cuda()


def test_distillation_two_steps_match_reference():
    from nf_distillation_amd.training import DistillationStep
    g = Golden("distill8")
    student = _build(g.cfg["student"], g.group("s_sd"))
    teacher = _build(g.cfg["teacher"], g.group("t_sd"))
    step = DistillationStep(student, teacher, nll_weight=0.9, kd_weight=0.1, lr=5e-4, clip_norm=30.0)
    assert step.s_idx == g.t("s_idx").tolist() and step.t_idx == g.t("t_idx").tolist()
    for i in range(2):
        p = f"step{i}/"
        x = g.t(p + "x").cuda()
        out = step.training_step(x, g.t(p + "noise_s").cuda(), g.t(p + "noise_t").cuda())
        torch.cuda.synchronize()
        # the teacher saw the student's noise plus its own (pl_module.py:216-225 + models/utils.py:38)
        assert torch.allclose(x.cpu(), g.t(p + "x") + g.t(p + "noise_s") + g.t(p + "noise_t"), atol=1e-7)
        assert rel_err(out["nll"].cpu(), g.t(p + "nll")) < 1e-5
        assert rel_err(out["kd"].cpu(), g.t(p + "kd")) < 1e-4
        assert rel_err(out["result_loss"].cpu(), g.t(p + "loss")) < 1e-5
        assert rel_err(step.opt.grad_norm().cpu(), g.t(p + "grad_norm")) < 1e-3
        sd = student.state_dict()
        for k, ref in g.group(p + "s_sd_after").items():
            assert float((sd[k].cpu() - ref).abs().max()) < 3e-5, (i, k)


def test_flat_adam_matches_torch_adam():
    """nfk_sumsq + nfk_adam_step against torch.optim.Adam + clip_grad_norm_ on random data."""
    from nf_distillation_amd.training import FlatAdam, FlatParams
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11)).cuda()
    ref.load_state_dict(model.state_dict())
    model.flow = torch.nn.Module()  # FlatAdam looks for model.flow._engine
    flat = FlatParams(model)
    opt = FlatAdam(flat, lr=1e-3, clip_norm=0.5, weight_decay=0.01)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=0.01)
    for it in range(3):
        grads = [torch.randn_like(p) for p in ref.parameters()]
        flat.zero_grad()
        for p, q, gr in zip(model.parameters(), ref.parameters(), grads):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        ropt.step()
        opt.step()
        assert rel_err(opt.grad_norm().cpu(), gn.cpu()) < 1e-5
        for p, q in zip(model.parameters(), ref.parameters()):
            assert float((p - q).abs().max()) < 1e-6
