"""The geometry bench.py times (CIFAR pair, B=64, parameters of seeds 11 / 12) against the oracle, step 0, both
precisions, through the eager step and through the default schedule (staged pipeline in one hipGraph; round 2's as well).  At B=64 the engine picks code
paths no smaller fixture reaches (level 1 as 256 unsplit workgroups, level 3 on 16-pixel tiles, weight-gradient split
caps, the pipelined schedule's split policy): this is where they meet the reference's arithmetic.  Bars: tests/oracle_check.py."""
import pytest
import torch

import oracle_check as OC

pytestmark = pytest.mark.gpu
BATCH = 64


@pytest.fixture(scope="module")
def oracle_ref():
    return OC.oracle_step(BATCH)


@pytest.mark.parametrize("dtype,schedule", [("f32", "eager"), ("f32", "staged_graph"), ("bf16", "eager"),
                                            ("bf16", "staged_graph"), ("bf16", "pipelined_graph")])
def test_bench_geometry_step_matches_oracle(oracle_ref, dtype, schedule):
    scal, grads, _ = OC.hip_step(dtype, schedule, BATCH)
    res = OC.compare(oracle_ref, scal, grads, dtype)
    print(dtype, schedule, res)
    assert res["ok"], (dtype, schedule, res)


def test_bench_geometry_merged_weight_gradient_tail_equals_unmerged_bf16():
    """with the slab reductions of a step riding on the NEXT step's merged launch, the first step of a
    level used to be reduced inside a launch whose chained coupling backward wrote the next level's layout into the same
    workspace (6 144 entries of a level-3 conv2 weight gradient at B=64).  The owed reductions are now flushed at level
    boundaries: merged and unmerged launches must give the same gradients, every tensor, at the bench geometry."""
    import nf_distillation_amd as nfa
    from param_util import randomize_state_dict, shapes_of, synthetic_images
    cfg = OC.cifar_cfg(8)
    x, noise = synthetic_images(BATCH, seed=321)
    res = {}
    for merge, chain in ((True, True), (False, True), (False, False)):
        m = nfa.create_glow_model(cfg)
        m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), OC.SEED_STUDENT))
        m = m.cuda().set_compute_dtype("bf16")
        eng = m.flow.engine()
        eng.merge_wgrad_tail, eng.chain_backward = merge, chain
        outs, bpd, _ = m(x.clone().cuda(), None, noise=noise.cuda())
        (bpd.mean() + 0.05 * sum(o.square().mean() for o in outs[::7])).backward()
        torch.cuda.synchronize()
        res[(merge, chain)] = {k: p.grad.clone() for k, p in m.named_parameters()}
    # merged vs unmerged launches of the chained backward: the same kernels' bodies on the same data -> bit for bit
    for k, v in res[(True, True)].items():
        w = res[(False, True)][k]
        assert torch.equal(v, w), ("merged", k, float((v - w).abs().max()), float(w.abs().max()))
    # chained vs per-step calls: the coupling tail's partial rows are one per channel-mix workgroup instead of one per
    # sample tile (another order of the same fp32 sums): 1e-6 of the tensor's largest entry
    base = res[(False, False)]
    for k, v in res[(True, True)].items():
        assert float((v - base[k]).abs().max()) <= 1e-6 * float(base[k].abs().max()) + 1e-12, ("chained", k)
