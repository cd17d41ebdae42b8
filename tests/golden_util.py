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
