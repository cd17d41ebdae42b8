"""Deterministic parameter randomisation shared by the golden generator and the tests.

Fresh reference models have all-zero Conv2dZeros / ActNorm parameters (models/layers.py:77-78,
251-252), so the coupling nets would output 0 and nothing would be exercised.  Every value is
re-drawn here from the CPU generator in state_dict order, as a function of (key, shape, seed)
only, so the same parameters can be rebuilt anywhere without shipping them.
"""
import math

import torch


def _gen(seed, idx):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed * 100003 + idx)
    return g


def randomize_state_dict(shapes, seed=0, dtype=torch.float32):
    """shapes: ordered {key: shape}.  Returns {key: tensor}."""
    out = {}
    for idx, (key, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        g = _gen(seed, idx)
        n = lambda s=1.0: torch.randn(shape, generator=g, dtype=torch.float64) * s
        leaf = key.split(".")[-1]
        if key.endswith("invconv.p"):
            c = shape[0]
            perm = torch.randperm(c, generator=g)
            v = torch.zeros(c, c, dtype=torch.float64)
            v[torch.arange(c), perm] = 1.0
        elif key.endswith("invconv.sign_s"):
            v = torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0).double()
        elif key.endswith("invconv.log_s"):
            v = n(0.1)
        elif key.endswith("invconv.lower") or key.endswith("invconv.upper"):
            v = n(0.3 / math.sqrt(shape[0]))
        elif key.endswith("invconv.weight"):
            q, _ = torch.linalg.qr(n())
            v = q + n(0.05)
        elif key == "prior_h":
            v = n(0.1)
        elif leaf == "logs":
            v = n(0.1)
        elif leaf == "bias":
            v = n(0.1)
        elif leaf == "weight":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if len(shape) == 4 and (".block.0." in key or ".block.2." in key):
                fan_out = shape[0] * shape[2] * shape[3]
                v = n(math.sqrt(2.0 / (fan_in + fan_out)))
            elif len(shape) == 4:
                v = n(0.5 / math.sqrt(fan_in))
            else:
                v = n(1.0 / math.sqrt(fan_in))
        else:
            raise KeyError(f"no randomisation rule for {key}")
        out[key] = v.to(dtype)
    return out


def shapes_of(sd):
    return {k: tuple(v.shape) for k, v in sd.items()}


def synthetic_images(batch, chw=(3, 32, 32), seed=42, dtype=torch.float32):
    """x = randint(0,256)/256 - 0.5 (data/src/utils.py:11-16 range) + U(0,1/256) noise."""
    g = _gen(seed, 7)
    x = torch.randint(0, 256, (batch,) + tuple(chw), generator=g).to(dtype) / 256.0 - 0.5
    noise = torch.rand((batch,) + tuple(chw), generator=g, dtype=torch.float64).to(dtype) / 256.0
    return x, noise
