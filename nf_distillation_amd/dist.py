"""Data-parallel plumbing for the distillation step: one process per GPU, RCCL over xGMI.

The path shards by independent samples: every rank holds a full replica of
teacher and student, takes B/N of the minibatch, and the only data-path collective is the
all-reduce of the student's gradients -- contiguous buckets of ONE flat fp32 arena, issued
(async, on RCCL's own stream) while the backward of earlier layers is still running.  The
teacher is frozen, so it never communicates.

Nothing here depends on the HIP kernels: it operates on a flat tensor through
``torch.distributed`` and is therefore testable on CPU with the gloo backend
(tests/test_ddp_gloo.py).
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """Contiguous, even split of the sample axis; remainder samples go to the low ranks."""
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def sample_noise(shape_chw, sample_ids, seed, device="cpu", dtype=torch.float32):
    """Dequantisation noise U(0,1/256) keyed by GLOBAL sample index, so that an N-rank run draws
    exactly what the 1-rank run draws for the same samples."""
    out = torch.empty((len(sample_ids),) + tuple(shape_chw), dtype=dtype)
    g = torch.Generator(device="cpu")
    for i, sid in enumerate(sample_ids):
        g.manual_seed(int(seed) * 1000003 + int(sid))
        out[i] = torch.rand(tuple(shape_chw), generator=g, dtype=torch.float64).to(dtype) / 256.0
    return out.to(device)


class GradAllReducer:
    """Bucketed all-reduce(sum) of a flat gradient arena.

    ``ready(lo)`` says "every element at offset >= lo is final" (backward runs from the last
    layer to the first, and the arena is laid out in layer order), and launches the buckets that
    have become complete.  ``finish()`` launches what is left and waits.  Averaging (1/N) is not
    done here: it is folded into the optimiser kernel's gradient scale.

    xGMI is point-to-point (7 links per GPU), so a few large buckets beat many small ones;
    the default 16 MiB gives 3 buckets for the 44 MB CIFAR student.
    """

    def __init__(self, flat_grad, group=None, bucket_bytes=16 << 20):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = flat_grad.numel()
        per = max(1, bucket_bytes // flat_grad.element_size())
        # buckets are cut from the END of the arena (first to become ready)
        self.bounds = []
        hi = n
        while hi > 0:
            lo = max(0, hi - per)
            self.bounds.append((lo, hi))
            hi = lo
        self.timing = None   # list of (start, end) event pairs around finish() when timing is on (bench.py: allreduce_ms)
        self.reset()

    def enable_timing(self, on=True):
        """Bracket every finish() (what is left of the bucketed all-reduce once the backward has been enqueued, plus the
        wait for all buckets) with events on the current stream; `timed_ms()` averages them."""
        self.timing = [] if on else None

    def timed_ms(self):
        if not self.timing:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.timing]
        return sum(ms) / len(ms)

    def reset(self):
        self.next = 0
        self.works = []

    def ready(self, lo):
        if self.world == 1:
            return
        while self.next < len(self.bounds) and self.bounds[self.next][0] >= lo:
            a, b = self.bounds[self.next]
            self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.next += 1

    def finish(self):
        if self.world == 1:
            return
        ev = None
        if self.timing is not None and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.ready(0)
        for w in self.works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        self.reset()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
