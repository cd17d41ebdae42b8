"""Software-pipelined step (teacher(t+1) || student(t)) against the sequential step: same noise draws in the same
order, so parameters after N batches must be bit-identical."""
import pytest
import torch

from golden_util import Golden

pytestmark = pytest.mark.gpu


def _make(g):
    import nf_distillation_amd as nfa
    from nf_distillation_amd.training import DistillationStep
    s = nfa.create_glow_model(dict(g.cfg["student"]))
    s.load_state_dict(g.group("s_sd"))
    t = nfa.create_glow_model(dict(g.cfg["teacher"]))
    t.load_state_dict(g.group("t_sd"))
    return DistillationStep(s.cuda(), t.cuda())


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipelined_equals_sequential(use_graph):
    from nf_distillation_amd.pipeline import PipelinedStep
    g = Golden("distill8")
    batches = [g.t("step0/x").cuda(), g.t("step1/x").cuda()]
    seq_n = 7
    xs = [batches[i % 2] for i in range(seq_n)]

    seq = _make(g)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    losses_seq = []
    for x in xs:
        n_s = torch.empty_like(x).uniform_(0, 1.0 / 256)
        n_t = torch.empty_like(x).uniform_(0, 1.0 / 256)
        losses_seq.append(float(seq.training_step(x.clone(), n_s, n_t)["result_loss"]))

    pip = _make(g)
    pip.enable_concurrency()
    runner = PipelinedStep(pip, xs[0], use_graph=use_graph, warmup=2)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    runner.prime(xs[0])
    losses_pip = []
    for x in xs[1:]:
        losses_pip.append(float(runner.training_step(x)["result_loss"]))
    losses_pip.append(float(runner.drain()["result_loss"]))
    torch.cuda.synchronize()
    if use_graph:
        assert runner.graph is not None, "capture did not happen"
        # graph replays draw from the graph-safe generator state: same distribution, different stream; check training
        assert all(abs(a - b) < 0.05 for a, b in zip(losses_pip, losses_seq))
        return
    assert losses_pip == losses_seq
    for (k, p), (_, q) in zip(pip.student.state_dict().items(), seq.student.state_dict().items()):
        assert torch.equal(p.cpu(), q.cpu()), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_teacher_two_batches_ahead_equals_sequential(use_graph):
    """AheadPipelinedStep (teacher(t+2) launched under the all-reduce / optimiser of batch t, three graphs on two streams):
    same noise draws in the same order as the sequential step, so losses and parameters after N batches are bit-identical
    in eager mode; replayed graphs draw from the graph-safe generator (same check as above: training tracks)."""
    from nf_distillation_amd.pipeline import AheadPipelinedStep
    g = Golden("distill8")
    batches = [g.t("step0/x").cuda(), g.t("step1/x").cuda()]
    seq_n = 8
    xs = [batches[i % 2] for i in range(seq_n)]

    seq = _make(g)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    losses_seq = []
    for x in xs:
        n_s = torch.empty_like(x).uniform_(0, 1.0 / 256)
        n_t = torch.empty_like(x).uniform_(0, 1.0 / 256)
        losses_seq.append(float(seq.training_step(x.clone(), n_s, n_t)["result_loss"]))

    pip = _make(g)
    pip.enable_concurrency()
    runner = AheadPipelinedStep(pip, xs[0], use_graph=use_graph, warmup=2)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    runner.prime(xs[0], xs[1])
    losses_pip = []
    for x in xs[2:]:
        losses_pip.append(float(runner.training_step(x)["result_loss"]))
    losses_pip += [float(o["result_loss"]) for o in runner.drain()]
    torch.cuda.synchronize()
    assert len(losses_pip) == seq_n
    if use_graph:
        assert runner.graph is not None and runner.graph_t is not None and runner.graph_b is not None, "capture did not happen"
        assert all(abs(a - b) < 0.05 for a, b in zip(losses_pip, losses_seq))
        return
    assert losses_pip == losses_seq
    for (k, p), (_, q) in zip(pip.student.state_dict().items(), seq.student.state_dict().items()):
        assert torch.equal(p.cpu(), q.cpu()), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_teacher_in_two_stages_equals_sequential(use_graph):
    """StagedPipelinedStep (teacher level 1 of batch t+2 || teacher deep levels of batch t+1 || student of batch t): the same
    noise draws in the same order as the sequential step -> identical losses and parameters in eager mode; replayed graphs
    draw from the graph-safe generator (training tracks)."""
    from nf_distillation_amd.pipeline import StagedPipelinedStep
    g = Golden("distill8")
    batches = [g.t("step0/x").cuda(), g.t("step1/x").cuda()]
    seq_n = 8
    xs = [batches[i % 2] for i in range(seq_n)]

    seq = _make(g)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    losses_seq = []
    for x in xs:
        n_s = torch.empty_like(x).uniform_(0, 1.0 / 256)
        n_t = torch.empty_like(x).uniform_(0, 1.0 / 256)
        losses_seq.append(float(seq.training_step(x.clone(), n_s, n_t)["result_loss"]))

    pip = _make(g)
    pip.enable_concurrency()
    runner = StagedPipelinedStep(pip, xs[0], use_graph=use_graph, warmup=2)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    runner.prime(xs[0], xs[1])
    losses_pip = []
    for x in xs[2:]:
        losses_pip.append(float(runner.training_step(x)["result_loss"]))
    losses_pip += [float(o["result_loss"]) for o in runner.drain()]
    torch.cuda.synchronize()
    assert len(losses_pip) == seq_n
    if use_graph:
        assert runner.graph is not None, "capture did not happen"
        assert all(abs(a - b) < 0.05 for a, b in zip(losses_pip, losses_seq))
        return
    assert losses_pip == losses_seq
    for (k, p), (_, q) in zip(pip.student.state_dict().items(), seq.student.state_dict().items()):
        assert torch.equal(p.cpu(), q.cpu()), k
