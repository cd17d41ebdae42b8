"""World-size-2 test of the data-parallel path on CPU (gloo): sharding + bucketed gradient
all-reduce of the flat arena + averaging reproduces the 1-rank gradient of the full batch.
The model is the CPU oracle (the HIP path cannot run here); the code under test is
nf_distillation_amd.dist, which is device-agnostic."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _grads(sd, cfg, x, noise, names):
    from oracle import glow_oracle as O
    for k in names:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    _, bpd, _ = O.glow_forward(sd, cfg, x.clone(), None, noise)
    bpd.mean().backward()
    return torch.cat([sd[k].grad.reshape(-1) for k in names])


def _worker(rank, world, port, out_path):
    from golden_util import Golden
    from nf_distillation_amd.dist import GradAllReducer, sample_noise, shard_range
    from oracle import glow_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    g = Golden("glow8")
    cfg, sd = g.cfg, g.group("sd")
    names = [k for k in sd if not O.is_buffer(k)]
    gen = torch.Generator().manual_seed(5)
    B = 6
    x = torch.randint(0, 256, (B, 3, 8, 8), generator=gen).float() / 256 - 0.5
    lo, hi = shard_range(B, rank, world)
    noise = sample_noise((3, 8, 8), range(lo, hi), seed=9)
    flat = _grads(sd, cfg, x[lo:hi], noise, names) * (hi - lo)   # sum over local samples
    red = GradAllReducer(flat, bucket_bytes=4096)
    assert len(red.bounds) > 3
    # backward finishes from the end of the arena: release it in three stages
    n = flat.numel()
    red.ready(2 * n // 3)
    red.ready(n // 3)
    red.finish()
    flat /= B
    if rank == 0:
        full_noise = sample_noise((3, 8, 8), range(B), seed=9)
        ref = _grads(sd, cfg, x, full_noise, names)
        torch.save({"got": flat, "ref": ref}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_one_rank(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    err = float((res["got"] - res["ref"]).abs().max() / res["ref"].abs().max())
    assert err < 1e-4, err


def test_shard_range_covers_batch():
    from nf_distillation_amd.dist import shard_range
    for B in (1, 7, 64, 513):
        for W in (1, 2, 3, 8):
            spans = [shard_range(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_noise_is_keyed_by_global_sample_index():
    from nf_distillation_amd.dist import sample_noise
    full = sample_noise((3, 4, 4), range(8), seed=1)
    part = sample_noise((3, 4, 4), range(4, 8), seed=1)
    assert torch.equal(full[4:], part) and float(full.max()) < 1 / 256 and float(full.min()) >= 0
