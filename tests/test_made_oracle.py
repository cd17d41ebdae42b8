"""The MADE / MAF block's CPU restatement (oracle/glow_oracle.py: made_step) checked against the mathematics it states --
there is NO reference code to pin it to (the reference's README mentions MAF, its source has no
MADE), so parity is UNPINNED; what can be checked is that the block is what Papamakarios et al. 2017 define: an
autoregressive bijection whose log|det J| is -sum alpha and whose inverse inverts it."""
import torch

from oracle import glow_oracle as O


def _sd(d, hid, ncond, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=0.5: torch.randn(*s, generator=g, dtype=torch.float64) * sc
    sd = {"actnorm.bias": r(1, d, sc=0.2), "actnorm.logs": r(1, d, sc=0.2)}
    dims = [(hid, d + ncond)] + [(hid, hid)] * 4 + [(2 * d, hid)]
    for i, (o, c) in zip((0, 2, 4, 6, 8, 10), dims):
        sd[f"block.{i}.weight"] = r(o, c, sc=0.6 / c ** 0.5)
        sd[f"block.{i}.bias"] = r(o, sc=0.1)
    return sd


def test_made_step_is_autoregressive_with_the_stated_logdet_and_inverse():
    for d, hid, ncond in ((6, 32, 0), (5, 16, 3), (9, 8, 0)):
        cfg = dict(is_1d=True, flow_coupling="made", y_condition=bool(ncond), y_classes=ncond)
        sd = _sd(d, hid, ncond, seed=d)
        x = torch.randn(3, d, dtype=torch.float64)
        y = torch.randn(3, ncond, dtype=torch.float64) if ncond else None
        ld0 = torch.zeros(3, dtype=torch.float64)
        u, ld = O.flow_step(sd, cfg, "", x, ld0, y, False)
        for b in range(3):
            yb = None if y is None else y[b:b + 1]
            J = torch.autograd.functional.jacobian(lambda t: O.flow_step(sd, cfg, "", t[None], ld0[:1], yb, False)[0][0], x[b])
            # the block reverses the feature order first: in the reversed input order the Jacobian is lower triangular
            Jr = J[:, torch.arange(d - 1, -1, -1)]
            assert float(torch.triu(Jr, 1).abs().max()) < 1e-12
            assert abs(float(torch.linalg.slogdet(J)[1] - ld[b])) < 1e-9
        xr, ldr = O.flow_step(sd, cfg, "", u, ld, y, True)
        assert float((xr - x).abs().max()) < 1e-9 and float(ldr.abs().max()) < 1e-9
