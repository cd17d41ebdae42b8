"""Host side of the 1-D (tabular) FlowStep: binds one `nfk_tab_step` per layer.

Reference: FlowStep with is_1d (models/flows.py:142-202), get_block_1d (models/flows.py:37-52),
ActNorm1d / InvertibleConv1x1 1-D branch (models/layers.py:159-170, 410-411)."""
import ctypes as C

import torch

from . import capi
from .capi import TabGrads, TabStepC, check, ptr

_LIN = (0, 2, 4, 6, 8, 10)


class TabStep:
    def __init__(self, rec, B, prep_base, off):
        layer = rec.mod
        self.layer, self.rec, self.B = layer, rec, B
        s = TabStepC()
        s.B, s.d, s.hid, s.ncond = B, rec.C, rec.hid, layer.condition_features
        s.additive = rec.additive
        # MADE block (modules.MADEStep): the kernels get W * mask (refreshed when a weight changes) and leave the gradient
        # wrt the masked weight in a buffer that is multiplied with the mask on its way into .grad
        self.made = getattr(layer, "flow_coupling", "") == "made"
        s.made = 1 if self.made else 0
        self.masked, self.masked_ver, self.gtmp = None, None, None
        for k, li in enumerate(_LIN):
            lin = layer.block[li]
            for t in (lin.weight, lin.bias):
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError("nfk tab kernels need contiguous fp32 parameters")
            s.W[k], s.b[k] = ptr(lin.weight), ptr(lin.bias)
        if self.made:
            self.masked = [torch.empty_like(layer.block[li].weight) for li in _LIN]
            self.gtmp = [torch.empty_like(layer.block[li].weight) for li in _LIN]
            for k in range(6):
                s.W[k] = ptr(self.masked[k])
        for nm in ("Wf", "cf", "Wi", "ci", "dld"):
            setattr(s, nm, prep_base + 4 * off[nm])
        self.cs = s
        lib = capi.lib()
        self.nblk = lib.nfk_tab_bwd_blocks(C.byref(s))
        self.nparam = lib.nfk_tab_param_floats(C.byref(s))
        self.part_mix = None       # partial (gW, gc) rows of the last backward; None = this step's backward has not run
        self.part_mix_inv = None
        self._bufs = {}            # persistent workspaces (stable pointers: descriptors are uploaded once, and a captured
                                   # hipGraph replays against the same addresses)

    def _refresh_masked(self):
        if not self.made:
            return
        ws = [self.layer.block[li].weight for li in _LIN]
        ver = tuple((w._version, w.data_ptr()) for w in ws)
        if ver != self.masked_ver:
            with torch.no_grad():
                for k, (w, m) in enumerate(zip(ws, self.layer.masks())):
                    torch.mul(w, m, out=self.masked[k])
            self.masked_ver = ver

    def mark_dirty(self):
        self.masked_ver = None

    def _buf(self, name, numel, device):
        t = self._bufs.get(name)
        if t is None or t.numel() != numel or t.device != device:
            t = self._bufs[name] = torch.empty(numel, device=device, dtype=torch.float32)
        return t

    def _cond(self, y_onehot, forward):
        # forward conditions on "y given", reverse on "model has condition features" (models/flows.py:152 vs :177)
        use = (y_onehot is not None) if forward else bool(self.cs.ncond)
        if use:
            if y_onehot is None or y_onehot.shape[1] != self.cs.ncond:
                raise RuntimeError(f"FlowStep expects {self.cs.ncond} condition features")
            return y_onehot.to(torch.float32).contiguous()
        if self.cs.ncond:
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (conditional block called without y_onehot)")
        return None

    def forward(self, x, ld, y_onehot, need):
        lib, s = capi.lib(), capi.stream()
        cond = self._cond(y_onehot, True)
        self._refresh_masked()
        out = torch.empty_like(x)
        ld_out = torch.empty_like(ld)
        check(lib.nfk_tab_fwd(C.byref(self.cs), ptr(x), ptr(cond), ptr(ld), ptr(out), ptr(ld_out), s), "nfk_tab_fwd")
        return out, ld_out, ((x, cond) if need else None)

    def inverse(self, y, y_onehot, need=False):
        lib, s = capi.lib(), capi.stream()
        cond = self._cond(y_onehot, False)
        self._refresh_masked()
        if self.made and need:
            raise NotImplementedError("the reverse pass with gradients is implemented for coupling steps only")
        out = torch.empty_like(y)
        check(lib.nfk_tab_inv(C.byref(self.cs), ptr(y), ptr(cond), None, ptr(out), None, s), "nfk_tab_inv")
        return (out, (y, cond)) if need else out

    def inverse_backward(self, saved, g_x, want_gcond):
        """Backward of `inverse` (reverse pass with gradients, pl_module.py:229-243): g_x = dL/d(output) -> g_y."""
        lib, s = capi.lib(), capi.stream()
        y, cond = saved
        blk = self.layer.block
        g = TabGrads()
        for k, li in enumerate(_LIN):
            g.gW[k], g.gb[k] = ptr(blk[li].weight.grad), ptr(blk[li].bias.grad)
        d = self.cs.d
        part = self._buf("part_inv", self.nblk * self.nparam, y.device)
        self.part_mix_inv = self._buf("mix_inv", self.nblk * (d * d + d), y.device)
        gy = torch.empty_like(y)
        gcond = torch.zeros_like(cond) if (cond is not None and want_gcond) else None
        check(lib.nfk_tab_inv_bwd(C.byref(self.cs), C.byref(g), ptr(y), ptr(cond), ptr(g_x), None, ptr(gy), ptr(gcond),
                                  ptr(part), ptr(self.part_mix_inv), s), "nfk_tab_inv_bwd")
        self._keep_inv = part
        return gy, gcond

    def backward(self, saved, g_out, g_ld, want_gcond):
        lib, s = capi.lib(), capi.stream()
        x, cond = saved
        blk = self.layer.block
        self._refresh_masked()
        g = TabGrads()
        for k, li in enumerate(_LIN):
            g.gW[k], g.gb[k] = ptr(blk[li].weight.grad), ptr(blk[li].bias.grad)
            if self.made:   # gradient wrt the MASKED weight: through the mask into .grad after the launch
                self.gtmp[k].zero_()
                g.gW[k] = ptr(self.gtmp[k])
        d = self.cs.d
        part = self._buf("part", self.nblk * self.nparam, x.device)
        self.part_mix = self._buf("mix", self.nblk * (d * d + d), x.device)
        gx = torch.empty_like(x)
        gcond = torch.zeros_like(cond) if (cond is not None and want_gcond) else None
        check(lib.nfk_tab_bwd(C.byref(self.cs), C.byref(g), ptr(x), ptr(cond), ptr(g_out), ptr(g_ld), ptr(gx), ptr(gcond),
                              ptr(part), ptr(self.part_mix), s), "nfk_tab_bwd")
        if self.made:
            for k, (li, m) in enumerate(zip(_LIN, self.layer.masks())):
                blk[li].weight.grad.addcmul_(self.gtmp[k], m)
        self._keep = part
        return gx, gcond
