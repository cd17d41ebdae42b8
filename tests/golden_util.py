"""Loader for tests/golden/*.npz (plain numpy arrays, allow_pickle=False)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
            self.arrays = {k: f[k] for k in f.files}
        self.cfg = json.loads(bytes(self.arrays.pop("cfg")).decode())

    def t(self, key, dtype=None):
        v = torch.from_numpy(np.array(self.arrays[key]))
        return v if dtype is None or not v.is_floating_point() else v.to(dtype)

    def has(self, key):
        return key in self.arrays

    def group(self, prefix, dtype=None):
        """{'a/b': ...} for every key under 'prefix/' (prefix stripped), in file order."""
        n = len(prefix) + 1
        return {k[n:]: self.t(k, dtype) for k in self.arrays if k.startswith(prefix + "/")}

    def seq(self, prefix, dtype=None):
        g = self.group(prefix, dtype)
        return [g[str(i)] for i in range(len(g))]


def rel_err(a, b):
    a, b = a.detach().double(), b.detach().double()
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ---- digests of large tensors (full-size fixtures keep these instead of the tensors) ---------------------------------
N_SAMPLES = 64


def sample_index(numel, key_idx):
    """Flat indices of the (at most) 64 entries of a tensor kept in a digest: a function of
    (numel, position of the key) only, so generator and tests agree."""
    g = torch.Generator().manual_seed(977 + 31 * key_idx)
    return torch.randint(0, numel, (min(N_SAMPLES, numel),), generator=g)


def tensor_digest(arrays, prefix, named):
    """abs-sum, sum (float64) and sampled entries of every tensor in `named` (ordered (key, tensor) pairs)."""
    abss, sums, samples = [], [], []
    for i, (_, v) in enumerate(named):
        v = v.detach().double().flatten().cpu()
        abss.append(v.abs().sum())
        sums.append(v.sum())
        smp = torch.zeros(N_SAMPLES, dtype=torch.float64)
        idx = sample_index(v.numel(), i)
        smp[: idx.numel()] = v[idx]
        samples.append(smp)
    arrays[prefix + "abssum"] = torch.stack(abss)
    arrays[prefix + "sum"] = torch.stack(sums)
    arrays[prefix + "samples"] = torch.stack(samples).float()


def digest_errors(golden, prefix, named):
    """Per tensor: (key, rel. error of the abs-sum, max |sample error| / max |reference sample|, sum error / abs-sum).
    Tensors whose reference abs-sum is 0 report the absolute values instead."""
    got = {}
    tensor_digest(got, "", named)
    ref_abs, ref_sum, ref_smp = golden.t(prefix + "abssum"), golden.t(prefix + "sum"), golden.t(prefix + "samples").double()
    out = []
    for i, (k, _) in enumerate(named):
        a = float(ref_abs[i])
        if a == 0.0:
            out.append((k, float(got["abssum"][i]), float(got["samples"][i].abs().max()), 0.0))
            continue
        smax = float(ref_smp[i].abs().max()) or a
        out.append((k, abs(float(got["abssum"][i]) - a) / a,
                    float((got["samples"][i].double() - ref_smp[i]).abs().max()) / smax,
                    abs(float(got["sum"][i]) - float(ref_sum[i])) / a))
    return out
