"""Two ranks sharing the one GPU of the test box (gloo transport for the collective, the real HIP
kernels for everything else): after one distillation step on half-batches each, both ranks hold
exactly the parameters a single rank gets from the full batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _models(g):
    import nf_distillation_amd as nfa
    s = nfa.create_glow_model(dict(g.cfg["student"]))
    s.load_state_dict(g.group("s_sd"))
    t = nfa.create_glow_model(dict(g.cfg["teacher"]))
    t.load_state_dict(g.group("t_sd"))
    return s.cuda(), t.cuda()


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from golden_util import Golden
    from nf_distillation_amd.dist import shard_range
    from nf_distillation_amd.training import DistillationStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = Golden("distill8")
    student, teacher = _models(g)
    step = DistillationStep(student, teacher, bucket_bytes=8192)
    step.enable_concurrency()
    assert step.world == world and len(step.reducer.bounds) > 2
    x, ns, nt = g.t("step0/x"), g.t("step0/noise_s"), g.t("step0/noise_t")
    lo, hi = shard_range(x.shape[0], rank, world)
    runner = step
    if os.environ.get("NFK_TEST_GRAPH") == "1":
        from nf_distillation_amd.training import CapturedStep
        runner = CapturedStep(step, x[lo:hi].cuda(), warmup=1, with_noise=True)
    xs, nss, nts = x[lo:hi].clone().cuda(), ns[lo:hi].cuda(), nt[lo:hi].cuda()
    if os.environ.get("NFK_TEST_GRAPH") == "2":   # software-pipelined step, split capture around the all-reduce
        from nf_distillation_amd.pipeline import PipelinedStep
        runner = PipelinedStep(step, xs, warmup=1, with_noise=True)
        runner.prime(xs, nss, nts)
        for _ in range(3):
            runner.training_step(xs, nss, nts)
        runner.drain()
    elif os.environ.get("NFK_TEST_GRAPH") == "3":   # teacher two batches ahead: its forward runs under the all-reduce
        from nf_distillation_amd.pipeline import AheadPipelinedStep
        runner = AheadPipelinedStep(step, xs, warmup=1, with_noise=True)
        runner.prime(xs, xs, nss, nts)
        for _ in range(2):
            runner.training_step(xs, nss, nts)
        runner.drain()
        assert runner.graph_t is not None, "the teacher branch was not captured"
    else:
        for _ in range(4):
            runner.training_step(xs.clone(), nss, nts)
    if runner is not step:
        assert runner.graph is not None and runner.graph_b is not None, "split capture did not happen"
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in student.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph", ["eager", "graph", "pipelined_graph", "teacher_ahead_graph"])
def test_two_ranks_equal_one_rank(tmp_path, graph):
    from golden_util import Golden
    from nf_distillation_amd.training import DistillationStep
    mode = {"eager": 0, "graph": 1, "pipelined_graph": 2, "teacher_ahead_graph": 3}[graph]
    port = 29700 + os.getpid() % 1000 + 7 * mode
    os.environ["NFK_TEST_GRAPH"] = str(mode)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = Golden("distill8")
    student, teacher = _models(g)
    step = DistillationStep(student, teacher)
    x, ns, nt = g.t("step0/x"), g.t("step0/noise_s"), g.t("step0/noise_t")
    for _ in range(4):
        step.training_step(x.clone().cuda(), ns.cuda(), nt.cuda())
    ref = {k: v.cpu() for k, v in student.state_dict().items()}
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for k in ref:
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}"
        # mean over the batch == mean of the two half-batch means; sums in a different order
        assert float((r0[k] - ref[k]).abs().max()) < 5e-6, k
