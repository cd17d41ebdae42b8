"""Split2d (models/layers.py:293-313) through the C ABI: the one-launch fused forward / backward (nfk_split.hip) against the
oracle's autograd, and against the generic GEMM path it replaces, on every tile geometry (rows + halo, whole image, several
images per tile, ragged last tile).  Tolerance: fp32 parity, 2e-5 relative to the tensor's scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu

GEOMS = [  # B, C, H, W
    (5, 12, 16, 16),   # CIFAR level 1: 4-row tiles with halo rows
    (3, 24, 8, 8),     # CIFAR level 2: one image per tile
    (7, 48, 4, 4),     # four images per tile, ragged last tile
    (2, 4, 32, 32),    # 2-row tiles
    (9, 8, 2, 2),      # sixteen images per tile
    (2, 64, 8, 8),     # 124 KB of LDS: close to the widest channel count the fused path takes
]


def _close(a, b, tol=2e-5):
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def _layer(C, dev):
    import nf_distillation_amd as nfa
    torch.manual_seed(C)
    lay = nfa.Split2d(num_channels=C)
    with torch.no_grad():   # Conv2dZeros starts at zero: give it something to compute
        lay.conv.conv.weight.normal_(0, 0.05)
        lay.conv.conv.bias.normal_(0, 0.1)
        lay.conv.logs.normal_(0, 0.05)
    return lay.to(dev)


def _run(lay, x, ld, gz, gl):
    for p in lay.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    ll = ld.clone().requires_grad_(True)
    z1, ldo = lay(xx, ll)
    ((z1 * gz).sum() + (ldo * gl).sum()).backward()
    return [z1.detach(), ldo.detach(), xx.grad, ll.grad] + [p.grad.clone() for p in lay.parameters()]


@pytest.mark.parametrize("B,C,H,W", GEOMS)
def test_split2d_fused_matches_oracle_and_gemm_path(B, C, H, W):
    import ctypes
    from nf_distillation_amd import capi
    from oracle import glow_oracle as O
    dev = torch.device("cuda:0")
    lay = _layer(C, dev)
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(B, C, H, W, generator=g)
    ld = torch.randn(B, generator=g)
    gz = torch.randn(B, C // 2, H, W, generator=g)
    gl = torch.randn(B, generator=g)
    lib = capi.lib()
    cs = capi.Split2d()
    cs.B, cs.H, cs.W, cs.C = B, H, W, C
    cs.w = capi.ptr(lay.conv.conv.weight)
    assert lib.nfk_split2d_fused_ok(ctypes.byref(cs)) == 1
    try:
        fused = _run(lay, x.to(dev), ld.to(dev), gz.to(dev), gl.to(dev))
        again = _run(lay, x.to(dev), ld.to(dev), gz.to(dev), gl.to(dev))
        lib.nfk_tune(capi.TUNE_SPLIT_FUSED, 0)
        assert lib.nfk_split2d_fused_ok(ctypes.byref(cs)) == 0
        gemm = _run(lay, x.to(dev), ld.to(dev), gz.to(dev), gl.to(dev))
    finally:
        lib.nfk_tune(capi.TUNE_SPLIT_FUSED, 1)
    for a, b in zip(fused, again):   # fixed-order reductions: bitwise reproducible
        assert torch.equal(a, b)
    # oracle (CPU, autograd)
    sd = {"conv." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in lay.conv.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    lo = ld.clone().requires_grad_(True)
    z1, ldo = O.split2d(sd, "", xo, lo, False)
    ((z1 * gz).sum() + (ldo * gl).sum()).backward()
    want = [z1.detach(), ldo.detach(), xo.grad, lo.grad] + [sd["conv." + k].grad for k, _ in lay.conv.named_parameters()]
    for a, b, c in zip(fused, gemm, want):
        _close(a.cpu(), c)
        _close(a.cpu(), b.cpu())


def test_split2d_fused_not_taken_for_unsupported_geometry():
    import ctypes
    from nf_distillation_amd import capi
    lib = capi.lib()
    cs = capi.Split2d()
    cs.B, cs.H, cs.W, cs.C = 2, 12, 12, 12       # not a power of two: the GEMM path
    dummy = torch.zeros(1, device="cuda:0")
    cs.w = capi.ptr(dummy)
    assert lib.nfk_split2d_fused_ok(ctypes.byref(cs)) == 0
    cs.H, cs.W, cs.C = 8, 8, 96                  # tiles + weights would not fit in LDS
    assert lib.nfk_split2d_fused_ok(ctypes.byref(cs)) == 0
    cs.H, cs.W, cs.C = 16, 16, 12
    assert lib.nfk_split2d_fused_ok(ctypes.byref(cs)) == 1
    cs.w = None                                   # no raw weight bound: the GEMM path
    assert lib.nfk_split2d_fused_ok(ctypes.byref(cs)) == 0
