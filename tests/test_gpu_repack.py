"""bf16 weight repack (nfk_repack_batch_bf16) against the three layouts spelled out with tensor ops.

The layouts replace what the reference hands to cuDNN as `conv.weight` (models/layers.py:170-228, 231-260):
  wt [Cout][Kp]      forward implicit GEMM, k = tap * Cin + ci, zero padded to a multiple of 64
  wd [Cin][Kpd]      dgrad: k = tap * Cout + co with the taps flipped
  wf                 wt once more in MFMA A-fragment order (1: 32x32x16, 2: 16x16x32; include/nfk.h)
  wg                 wd once more in MFMA A-fragment order (the fused step backward's operands; rows behind Cin of a
                     16-row layout are left untouched: the caller keeps the buffer zeroed)
bf16 conversion is round-to-nearest-even, so every element must agree bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (Cout, Cin, taps, frag): the shapes of Glow's coupling nets at CIFAR levels 1-3 and CelebA level 4, small hidden
# sizes, and shapes that no tile scheme divides
CASES = [(512, 512, 1, 1), (512, 6, 9, 1), (512, 12, 9, 1), (512, 24, 9, 1), (512, 48, 9, 1),
         (12, 512, 9, 2), (24, 512, 9, 2), (48, 512, 9, 2), (96, 512, 9, 2), (6, 512, 9, 2),
         (32, 32, 1, 1), (32, 6, 9, 1), (12, 32, 9, 2), (64, 40, 1, 1), (20, 10, 9, 0), (7, 13, 9, 2), (33, 70, 1, 0), (56, 64, 9, 2), (56, 64, 9, 0), (512, 72, 9, 1)]


def _expected(w, frag, Kp, Kpd):
    Cout, Cin, taps = w.shape
    wb = w.to(torch.bfloat16)
    wt = torch.zeros(Cout, Kp, dtype=torch.bfloat16)
    wt[:, :taps * Cin] = wb.permute(0, 2, 1).reshape(Cout, taps * Cin)          # k = tap*Cin + ci
    wd = torch.zeros(Cin, Kpd, dtype=torch.bfloat16)
    wd[:, :taps * Cout] = wb.flip(2).permute(1, 2, 0).reshape(Cin, taps * Cout)  # k = tap'*Cout + co, taps flipped
    wf = None
    if frag:
        rows, kk = (32, 16) if frag == 1 else (16, 32)
        R = (Cout + rows - 1) // rows * rows
        full = torch.zeros(R, Kp, dtype=torch.bfloat16)
        full[:Cout] = wt
        # fragment (nt, ks): lane l holds row nt*rows + l % rows, k = ks*kk + (l // rows)*8 .. +8
        f = full.view(R // rows, rows, Kp // kk, kk // 8, 8).permute(0, 2, 3, 1, 4)   # [nt][ks][khalf][row][8]
        wf = f.reshape(-1).contiguous()
    return wt, wd, wf


def _frag_order(mat, kind):
    """[rows][Kp] -> fragment order of `kind` (rows padded with zeros to the fragment height)."""
    rows, kk = (32, 16) if kind == 1 else (16, 32)
    R = (mat.shape[0] + rows - 1) // rows * rows
    full = torch.zeros(R, mat.shape[1], dtype=mat.dtype)
    full[:mat.shape[0]] = mat
    return full.view(R // rows, rows, mat.shape[1] // kk, kk // 8, 8).permute(0, 2, 3, 1, 4).reshape(-1).contiguous()


def _gkind(Cout, Cin, taps):
    """dgrad fragment layout requested for a case: 32-row fragments where Cin allows, else 16-row ones (K % 32 == 0)."""
    return 1 if Cin % 32 == 0 else 2


@pytest.mark.parametrize("entry", ["nfk_repack_batch_bf16", "nfk_prep_repack"])
def test_repack_bf16_layouts(entry):
    """`nfk_prep_repack` runs the same repack bodies inside the launch that also prepares the steps (two 256-thread
    workers per workgroup there); called here with no steps and no f32 weights."""
    from nf_distillation_amd import capi
    lib = capi.lib()
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(3)
    descs, keep, exp = [], [], []
    for Cout, Cin, taps, frag in CASES:
        w = torch.randn(Cout, Cin, taps, generator=gen)
        Kp, Kpd = lib.nfk_padded_k_bf16(taps * Cin), lib.nfk_padded_k_bf16(taps * Cout)
        ewt, ewd, ewf = _expected(w, frag, Kp, Kpd)
        wdev = w.to(dev)
        wt = torch.full((Cout * Kp,), 7.0, dtype=torch.bfloat16, device=dev)     # poison: padding must be written
        wd = torch.full((Cin * Kpd,), 7.0, dtype=torch.bfloat16, device=dev)
        wf = torch.full((ewf.numel(),), 7.0, dtype=torch.bfloat16, device=dev) if frag else None
        d = capi.RepackDesc()
        d.w, d.wt, d.wd = wdev.data_ptr(), wt.data_ptr(), wd.data_ptr()
        d.Cout, d.Cin, d.taps, d.frag = Cout, Cin, taps, frag
        d.wf = wf.data_ptr() if frag else None
        gk = _gkind(Cout, Cin, taps)
        wg = torch.zeros(_frag_order(ewd, gk).numel(), dtype=torch.bfloat16, device=dev)
        d.wg, d.gfrag = wg.data_ptr(), gk
        descs.append(d)
        keep.append((wdev, wt, wd, wf, wg))
        exp.append((ewt, ewd, ewf))
    dd = capi.struct_array_to_device(descs, dev)
    if entry == "nfk_repack_batch_bf16":
        capi.check(lib.nfk_repack_batch_bf16(dd.data_ptr(), len(descs), None), entry)
    else:
        capi.check(lib.nfk_prep_repack(None, 0, 0, None, 0, dd.data_ptr(), len(descs), None), entry)
    torch.cuda.synchronize()
    for case, (_, wt, wd, wf, wg), (ewt, ewd, ewf) in zip(CASES, keep, exp):
        ewg = _frag_order(ewd, _gkind(*case[:3]))
        assert torch.equal(wg.cpu().view(torch.int16), ewg.view(torch.int16)), ("wg", case)
        assert torch.equal(wt.cpu().view(torch.int16), ewt.reshape(-1).view(torch.int16)), ("wt", case)
        assert torch.equal(wd.cpu().view(torch.int16), ewd.reshape(-1).view(torch.int16)), ("wd", case)
        if wf is not None:
            assert torch.equal(wf.cpu().view(torch.int16), ewf.view(torch.int16)), ("wf", case)
