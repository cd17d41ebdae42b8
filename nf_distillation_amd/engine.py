"""FlowEngine: runs a list of flow layers (SqueezeLayer / FlowStep / Split2d) through libnfk.

One engine per ``FlowNet``.  It owns, per (batch, input shape, device):
  * the ctypes descriptors of every layer (raw parameter pointers, shapes),
  * the "prep" arena: per-step fused ActNorm x invconv matrices, log-det constants, exp(logs)
    vectors and the repacked conv weights -- produced by TWO launches for the whole model
    (``nfk_prep`` + ``nfk_repack_batch``) and cached while parameters are frozen (teacher),
  * nothing else: activations are allocated per call from torch's caching allocator.

Autograd: the whole encode is ONE ``torch.autograd.Function`` node.  Its backward walks the
layers in reverse through ``nfk_step2d_bwd`` / ``nfk_split2d_bwd`` and accumulates parameter
gradients straight into ``param.grad`` (pre-allocated, e.g. views of one flat buffer --
see training.FlatParams), so no per-parameter autograd bookkeeping runs on the hot path.
"""
import ctypes as C
import os

import torch

from . import capi
from .capi import PrepDesc, PrepGrad, RepackDesc, Split2d as CSplit2d, Step2d, Step2dGrads, check, ptr


def _a4(n):
    return (n + 3) & ~3


class _Rec:
    __slots__ = ("kind", "idx", "mod", "C", "H", "W", "hid", "Cout", "additive", "cs", "prep_i", "views",
                 "params", "tab")


def _nhwc(t):
    """[B,C,H,W] logical tensor -> dense [B,H,W,C] tensor (no copy if already channels-last)."""
    p = t.permute(0, 2, 3, 1)
    return p if p.is_contiguous() else p.contiguous()


def _check_param(p, name):
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise TypeError(f"{name}: nfk kernels need contiguous fp32 parameters")
    capi.require_cuda(p, name)


class FlowEngine:
    def __init__(self, layers):
        from . import modules as M
        self.M = M
        self.layers = layers
        self.state = None
        self.key = None
        self.is_1d = any(isinstance(l, (M.FlowStep, M.MADEStep)) and l.is_1d for l in layers)
        self._params = None
        self._tensor_list = None
        self.grad_hook = None  # callable(layer_index): that layer's parameter grads are final
        self.first_layer_hook = None  # callable(): first layer enqueued (input x may now be reused)
        self.fused = True      # bf16 mode: one fused launch per FlowStep forward where the geometry allows
        # hidden split of the fused step on small grids also for passes that keep nothing (a frozen teacher): on the
        # critical path it shortens the step; beside another chain (pipeline.py) it only takes CUs away from that chain
        self.fused_split_inference = 4   # ways: 4, 2, 0 (none) or "frugal" (see _EncodeFn.forward)
        self.fused_split_training = 4    # the same for a forward that keeps what the backward needs
        self.chain_backward = True       # nfk_step2d_bwd_chain (False: one independent nfk_step2d_bwd call per step)
        # fused backward: a step's weight-gradient GEMMs share ONE launch with the channel mix that ends the step; the slab
        # reductions that wait for them ride on the next step's launch (nfk_step2d_pending)
        self.merge_wgrad_tail = True
        # all FlowSteps of a level in ONE launch where the pixel tiles are whole images (nfk_run2d_fwd_fused; bit-identical
        # to the per-step launches without hidden split).  "inference": only for passes that keep nothing (a frozen teacher)
        self.fused_runs = True
        # bf16 fused backward per LEVEL (nfk_level2d_bwd): one launch per step (dgrad chain + channel mix + the coupling
        # backward of the step below) -- or ONE for all steps of a level whose tiles are whole images (`level_multi`) --, then
        # one launch with the level's 3 K weight-gradient GEMMs and one with all reductions.  False: the per-step calls
        # (nfk_step2d_bwd_chain), which the level form is tested against
        self.level_backward = os.environ.get("NFK_LEVEL_BACKWARD", "1") != "0"   # (environment knobs: diagnostics)
        self.level_multi = os.environ.get("NFK_LEVEL_MULTI", "1") != "0"
        # the per-level launches on the auxiliary stream (`aux`, set by training.enable_concurrency; without one they stay on
        # the caller's), joined once per backward: they then run beside the next level's step launches (+3 % of the step,
        # bit-identical: tests/test_gpu_level_bwd.py).  Opt-in (bench.py turns it on for its one-GPU headline leg): with it, a
        # captured PipelinedStep graph (caller's stream + teacher's side stream + this one) crashed inside hipGraphLaunch
        # (hip::Graph::UpdateStreams, the HIP runtime bundled with this PyTorch) -- only in a process that had captured and
        # destroyed other graphs with an auxiliary-stream branch before (tests/test_gpu_fullsize.py in file order; a fresh
        # process, other orders and the four-stream StagedPipelinedStep graph did not)
        self.level_wgrad_aux = os.environ.get("NFK_LEVEL_WGRAD_AUX", "0") == "1"
        # ... and, where a level's steps are one launch each, could go out for every `level_wgrad_chunk` steps instead of once
        # per level (0).  Measured and left off: beside the level-1 step launches, which want the whole chip themselves, the
        # chunks cost more than the shorter tail gains (CIFAR B=64: 20 330 images/s at 0, 20 040 / 19 620 / 18 670 at 4 / 2 / 1)
        self.level_wgrad_chunk = 0
        self.bwd_mode = capi.BWD_DEFAULT   # nfk_step2d_grads.bwd_mode: how the bf16 backward of a coupling net is launched
        self.defer_fused_wgrad = False   # bf16 fused backward: weight-gradient launches as a chain of their own on `aux`
        self.aux = None        # nfk_aux handle (training.enable_concurrency): auxiliary stream of the weight-gradient GEMMs
        self.aux_stream = None   # ... and that stream as a torch object (the deferred half of _refresh runs there)
        self.defer_refresh = os.environ.get("NFK_DEFER_REFRESH", "1") == "1"   # (see _refresh)
        self.bf16 = False      # throughput mode: bf16 operands / fp32 accumulation in the coupling-net convs

    # ------------------------------------------------------------------ parameters
    def params(self):
        if self._params is None:
            ps = []
            for l in self.layers:
                ps.extend(p for p in l.parameters())
            self._params = ps
        return self._params

    def _tensors(self):
        """Parameters and buffers of all layers.  Parameter objects are stable (their .data may be re-homed) and are
        listed once; buffers are REPLACED by `Module.to()` and by `inherit_permutation_matrix` (`invconv.p = ...`), so
        they are read from the modules' buffer dicts on every call."""
        if self._tensor_list is None:
            ts, holders = [], []
            for l in self.layers:
                ts.extend(l.parameters())
                holders.extend(m for m in l.modules() if m._buffers)
            self._tensor_list, self._buffer_holders = ts, holders
        bufs = [b for m in self._buffer_holders for b in m._buffers.values() if b is not None]
        return self._tensor_list + bufs

    def _signature(self):
        return tuple([t.data_ptr() for t in self._tensors()])

    def _version(self):
        return sum([t._version for t in self._tensors()])

    def set_compute_dtype(self, name):
        """"f32" (parity mode, default) or "bf16" (bf16 operands, fp32 accumulate in the coupling-net convs)."""
        if name not in ("f32", "bf16"):
            raise ValueError("compute dtype must be 'f32' or 'bf16'")
        want = name == "bf16"
        if want != self.bf16:
            self.bf16, self.state, self.key = want, None, None

    def mark_dirty(self):
        if self.state is not None:
            self.state["version"] = None
            for r in self.state["recs"]:
                if r.tab is not None:
                    r.tab.mark_dirty()   # (MADE blocks: W * mask is rebuilt)

    # ------------------------------------------------------------------ build
    def _build(self, B, in_shape, device):
        M = self.M
        lib = capi.lib()
        recs, prep_descs, repack_descs = [], [], []
        if self.is_1d:
            Cc, H, W = in_shape[0], 1, 1
        else:
            Cc, H, W = in_shape
        prep_floats = 0
        pack_floats = 0   # Split2d convs (always fp32)
        step_pack = 0     # FlowStep convs: elements of fp32 or bf16
        plan = []  # (rec, offsets)
        for i, layer in enumerate(self.layers):
            r = _Rec()
            r.idx, r.mod, r.prep_i, r.cs, r.tab = i, layer, -1, None, None
            if isinstance(layer, M.SqueezeLayer):
                r.kind, r.C, r.H, r.W = "squeeze", Cc, H, W  # INPUT dims
                Cc, H, W = Cc * 4, H // 2, W // 2
            elif isinstance(layer, (M.FlowStep, M.MADEStep)):
                if layer.in_channels != Cc:
                    raise ValueError(f"layer {i}: expected {layer.in_channels} channels, got {Cc}")
                r.kind = "tab" if layer.is_1d else "step"
                r.C, r.H, r.W, r.hid = Cc, H, W, layer.hidden_channels
                r.additive = 1 if layer.flow_coupling == "additive" else 0
                r.Cout = 2 * Cc if layer.flow_coupling == "made" else (Cc - Cc // 2) * (1 if r.additive else 2)
                off = {}
                o = prep_floats
                for name, n in (("Wf", Cc * Cc), ("cf", Cc), ("Wi", Cc * Cc), ("ci", Cc), ("dld", 4),
                                ("es1", 0 if layer.is_1d else r.hid), ("es2", 0 if layer.is_1d else r.hid),
                                ("es3", 0 if layer.is_1d else r.Cout), ("scratch", 5 * Cc * Cc)):
                    off[name] = o
                    o += _a4(n)
                prep_floats = o
                if not layer.is_1d:
                    Ch, hid = Cc // 2, r.hid
                    pk = lib.nfk_padded_k_bf16 if self.bf16 else lib.nfk_padded_k
                    if self.bf16 and hid % 8:
                        raise ValueError("bf16 mode needs hidden_channels % 8 == 0")
                    o = step_pack
                    for name, n in (("wt1", hid * pk(9 * Ch)), ("wd1", Ch * pk(9 * hid)),
                                    ("wt2", hid * pk(hid)), ("wd2", hid * pk(hid)),
                                    ("wt3", r.Cout * pk(9 * hid)), ("wd3", hid * pk(9 * r.Cout))):
                        off[name] = o
                        o += (n + 7) & ~7
                    if self.bf16 and hid % 32 == 0:
                        # fragment-ordered copies of the forward weights for the fused step kernel (nfk_fused2.hip)
                        for name, n in (("wf1", hid * pk(9 * Ch)), ("wf2", hid * pk(hid)),
                                        ("wf3", (r.Cout + 15) // 16 * 16 * pk(9 * hid))):
                            off[name] = o
                            o += (n + 7) & ~7
                        if any(p.requires_grad for p in layer.parameters()):
                            # ... and of the dgrad weights, for the fused step backward (nfk_fusedb.hip); rows behind
                            # C/2 of wg1 stay zero (the buffer is zero-initialised and the repack never writes them)
                            for name, n in (("wg3", hid * pk(9 * r.Cout)), ("wg2", hid * pk(hid)),
                                            ("wg1", (Ch + 15) // 16 * 16 * pk(9 * hid))):
                                off[name] = o
                                o += (n + 7) & ~7
                    step_pack = o
                plan.append((r, off))
            elif isinstance(layer, M.Split2d):
                r.kind, r.C, r.H, r.W = "split", Cc, H, W
                off = {"es3": prep_floats}
                prep_floats += _a4(Cc)
                Ch = Cc // 2
                off["wt"] = pack_floats
                pack_floats += _a4(Cc * lib.nfk_padded_k(9 * Ch))
                off["wd"] = pack_floats
                pack_floats += _a4(Ch * lib.nfk_padded_k(9 * Cc))
                plan.append((r, off))
                Cc = Cc // 2
            else:
                raise TypeError(f"unsupported layer type {type(layer).__name__}")
            recs.append(r)

        prep_buf = torch.zeros(max(prep_floats, 4), dtype=torch.float32, device=device)
        pack_buf = torch.zeros(max(pack_floats, 4), dtype=torch.float32, device=device)
        # +1024: the fused kernel's padded conv1 k-steps may read up to 512 B past a weight row (against zero columns)
        step_buf = torch.zeros(max(step_pack, 8) + 1024, dtype=torch.bfloat16 if self.bf16 else torch.float32,
                               device=device)
        pb, kb, sb, esz = prep_buf.data_ptr(), pack_buf.data_ptr(), step_buf.data_ptr(), (2 if self.bf16 else 4)
        step_repack = []
        keep = []  # tensors that must stay alive (perm indices, zero vectors)
        any_dense = 0
        for r, off in plan:
            layer = r.mod
            d = PrepDesc()
            if r.kind in ("step", "tab"):
                an = layer.actnorm
                _check_param(an.bias, "actnorm.bias")
                if r.C > 256:
                    raise ValueError("a FlowStep mixes at most 256 channels / features (prep kernels' LDS vectors)")
                d.C, d.pixels, d.is_1d = r.C, r.H * r.W, 1 if layer.is_1d else 0
                d.an_bias, d.an_logs = ptr(an.bias), ptr(an.logs)
                if layer.flow_permutation_type == "invconv":
                    ic = layer.invconv
                    if ic.LU_decomposed:
                        d.perm_kind = 0
                        for nm in ("lower", "upper", "log_s", "p", "sign_s"):
                            t = getattr(ic, nm)
                            _check_param(t, "invconv." + nm)
                            setattr(d, nm, ptr(t))
                    else:
                        d.perm_kind = 1
                        any_dense = 1
                        _check_param(ic.weight, "invconv.weight")
                        d.weight = ptr(ic.weight)
                else:
                    perm = layer.shuffle if layer.flow_permutation_type == "shuffle" else layer.reverse
                    idx = perm.indices.to(device=device, dtype=torch.int32).contiguous()
                    keep.append(idx)
                    d.perm_kind = 2
                    d.perm_idx = ptr(idx)
                for nm in ("Wf", "cf", "Wi", "ci", "dld", "scratch"):
                    setattr(d, nm, pb + 4 * off[nm])
                if r.kind == "step":
                    blk = layer.block
                    d.n1 = d.n2 = r.hid
                    d.n3 = r.Cout
                    d.l1, d.l2, d.l3 = ptr(blk[0].actnorm.logs), ptr(blk[2].actnorm.logs), ptr(blk[4].logs)
                    for nm in ("es1", "es2", "es3"):
                        setattr(d, nm, pb + 4 * off[nm])
                    cs = Step2d()
                    cs.B, cs.H, cs.W, cs.C, cs.hidden, cs.Cout, cs.additive = B, r.H, r.W, r.C, r.hid, r.Cout, r.additive
                    cs.bf16 = 1 if self.bf16 else 0
                    cs.b1, cs.b2, cs.bias3 = ptr(blk[0].actnorm.bias), ptr(blk[2].actnorm.bias), ptr(blk[4].conv.bias)
                    for nm in ("Wf", "cf", "Wi", "ci", "dld", "es1", "es2", "es3"):
                        setattr(cs, nm, pb + 4 * off[nm])
                    for nm in ("wt1", "wt2", "wt3", "wd1", "wd2", "wd3"):
                        setattr(cs, nm, sb + esz * off[nm])
                    for nm in ("wf1", "wf2", "wf3", "wg1", "wg2", "wg3"):
                        setattr(cs, nm, sb + esz * off[nm] if nm in off else None)
                    r.cs = cs
                    Ch = r.C // 2
                    for w, co, ci, taps, a, b, f, fk, gname, gk in (
                            (blk[0].conv.weight, r.hid, Ch, 9, "wt1", "wd1", "wf1", 1, "wg1", 2),
                            (blk[2].conv.weight, r.hid, r.hid, 1, "wt2", "wd2", "wf2", 1, "wg2", 1),
                            (blk[4].conv.weight, r.Cout, r.hid, 9, "wt3", "wd3", "wf3", 2, "wg3", 1)):
                        _check_param(w, "conv.weight")
                        if tuple(w.shape) != (co, ci, 3 if taps == 9 else 1, 3 if taps == 9 else 1):
                            raise ValueError(f"layer {r.idx}: unexpected conv weight shape {tuple(w.shape)}")
                        rd = RepackDesc()
                        rd.w, rd.wt, rd.wd = ptr(w), sb + esz * off[a], sb + esz * off[b]
                        rd.Cout, rd.Cin, rd.taps = co, ci, taps
                        if f in off:
                            rd.wf, rd.frag = sb + esz * off[f], fk
                        if gname in off:
                            rd.wg, rd.gfrag = sb + esz * off[gname], gk
                        step_repack.append(rd)
                else:
                    from .tabular import TabStep
                    r.tab = TabStep(r, B, pb, off)
            else:  # split
                cz = layer.conv
                d.C, d.pixels, d.n3, d.l3, d.es3 = 0, r.H * r.W, r.C, ptr(cz.logs), pb + 4 * off["es3"]
                cs = CSplit2d()
                cs.B, cs.H, cs.W, cs.C = B, r.H, r.W, r.C
                cs.bias, cs.es = ptr(cz.conv.bias), pb + 4 * off["es3"]
                cs.wt, cs.wd = kb + 4 * off["wt"], kb + 4 * off["wd"]
                cs.w = ptr(cz.conv.weight)
                r.cs = cs
                rd = RepackDesc()
                rd.w, rd.wt, rd.wd = ptr(cz.conv.weight), cs.wt, cs.wd
                rd.Cout, rd.Cin, rd.taps = r.C, r.C // 2, 9
                repack_descs.append(rd)
            r.prep_i = len(prep_descs)
            prep_descs.append(d)

        # runs of FlowSteps that one launch can walk (same geometry, whole-image tiles): device array of their descriptors
        runs = {}
        if self.bf16 and not self.is_1d:
            i = 0
            while i < len(recs):
                r = recs[i]
                if r.kind == "step" and r.cs.wf1 and lib.nfk_run2d_fused_ok(C.byref(r.cs)):
                    j = i
                    while (j < len(recs) and recs[j].kind == "step" and j - i < 64 and recs[j].cs.wf1 and
                           (recs[j].C, recs[j].H, recs[j].W, recs[j].hid, recs[j].Cout, recs[j].additive) ==
                           (r.C, r.H, r.W, r.hid, r.Cout, r.additive)):
                        j += 1
                    if j - i >= 2:
                        runs[i] = {"n": j - i, "dev": capi.struct_array_to_device([q.cs for q in recs[i:j]], device)}
                    i = j
                else:
                    i += 1

        # levels: maximal runs of FlowSteps of one geometry (the unit of nfk_level2d_bwd), keyed by their first layer index
        levels = {}
        if self.bf16 and not self.is_1d:
            i = 0
            while i < len(recs):
                r = recs[i]
                if r.kind != "step":
                    i += 1
                    continue
                j = i
                while (j < len(recs) and recs[j].kind == "step" and
                       (recs[j].C, recs[j].H, recs[j].W, recs[j].hid, recs[j].Cout, recs[j].additive) ==
                       (r.C, r.H, r.W, r.hid, r.Cout, r.additive)):
                    j += 1
                host = (Step2d * (j - i))(*[q.cs for q in recs[i:j]])
                ok = bool(r.cs.wg1) and all(lib.nfk_step2d_fused_bwd_ok(C.byref(q.cs)) for q in recs[i:j])
                levels[i] = {"n": j - i, "host": host, "bwd_ok": ok and bool(lib.nfk_level_bwd_ok(C.byref(r.cs))),
                             "dev": (runs[i]["dev"] if (i in runs and runs[i]["n"] == j - i)
                                     else capi.struct_array_to_device([q.cs for q in recs[i:j]], device))}
                i = j

        # every step on the fused kernels (forward and, where it trains, backward): they read the fragment-ordered weight
        # copies only, so the per-forward repack can leave the row-major ones out (the reverse pass and the per-conv
        # launches read those: _refresh repacks in full for them)
        steps_ = [q for q in recs if q.kind == "step"]
        all_fused = bool(self.bf16 and not self.is_1d and steps_) and all(
            q.cs.wf1 and lib.nfk_step2d_fused_tiles(C.byref(q.cs)) > 0 and
            (not any(p.requires_grad for p in q.mod.parameters()) or (q.cs.wg1 and lib.nfk_step2d_fused_bwd_ok(C.byref(q.cs))))
            for q in steps_)
        step_repack_frag_dev = None
        if all_fused:
            for rd in step_repack:
                rd.flags = capi.REPACK_FRAG_ONLY
            step_repack_frag_dev = capi.struct_array_to_device(step_repack, device)
            for rd in step_repack:
                rd.flags = 0

        # where the per-forward parameter products can be cut in two: everything the FIRST level reads (its steps' channel-mix
        # matrices and weight fragments) | the rest, which the forward needs only behind the first Split2d (_refresh)
        first_split = next((q for q in recs if q.kind == "split"), None)
        refresh_cut = None
        if first_split is not None and first_split.prep_i > 0 and all(q.kind == "step" for q in recs if q.idx < first_split.idx and q.kind != "squeeze"):
            n_lvl1 = sum(1 for q in recs if q.kind == "step" and q.idx < first_split.idx)
            if n_lvl1 == first_split.prep_i:
                refresh_cut = (first_split.prep_i, 3 * n_lvl1, first_split.idx)

        self.state = {
            "runs": runs, "levels": levels, "refresh_cut": refresh_cut, "refresh_event": None,
            "B": B, "recs": recs, "prep_buf": prep_buf, "pack_buf": pack_buf, "keep": keep,
            "prep_dev": capi.struct_array_to_device(prep_descs, device), "n_prep": len(prep_descs),
            "prep_host": prep_descs,
            "repack_dev": capi.struct_array_to_device(repack_descs, device), "n_repack": len(repack_descs),
            "step_repack_dev": capi.struct_array_to_device(step_repack, device), "n_step_repack": len(step_repack),
            "step_repack_frag_dev": step_repack_frag_dev, "has_plain": False,
            "step_buf": step_buf,
            "any_dense": any_dense, "version": None, "has_inverse": False, "sig": self._signature(),
            "out_shape": (Cc, H, W), "grad_sig": None, "grad_dev": None,
        }

    def _ensure(self, x):
        capi.require_cuda(x, "input")
        if x.dtype != torch.float32:
            raise TypeError("nfk kernels compute in fp32; got " + str(x.dtype))
        B = x.shape[0]
        in_shape = tuple(x.shape[1:])
        key = (B, in_shape, x.device)
        if self.state is None or self.key != key or self.state["sig"] != self._signature():
            self._build(B, in_shape, x.device)
            self.key = key
        return self.state

    def _refresh(self, st, want_inverse, may_defer=False):
        """Run prep (+ repack) if parameters may have changed since the cached products were made.
        may_defer (the encode pass): with the auxiliary stream in use (level_wgrad_aux), only what the first level reads is
        produced on the caller's stream; the rest goes out on the auxiliary stream and the pass waits for it in front of the
        first Split2d (st["refresh_event"]) -- the launch sits at the head of the training step's critical chain, and most of
        its time is the wide steps' LU products and weight fragments."""
        if st.get("refresh_event") is not None:   # (a pass that ended before its wait: never leave the products half-made)
            torch.cuda.current_stream().wait_event(st["refresh_event"])
            st["refresh_event"] = None
        trainable = any(p.requires_grad for p in self.params())
        ver = self._version()
        # row-major weight copies: read by the reverse pass and by every path but the fused forward / fused backward
        want_plain = bool(want_inverse or st["step_repack_frag_dev"] is None or not self.fused or
                          self.bwd_mode not in (capi.BWD_DEFAULT, capi.BWD_FUSED))
        fresh = ((not trainable) and st["version"] == ver and (st["has_inverse"] or not want_inverse) and
                 (st["has_plain"] or not want_plain))
        if fresh:
            return
        lib, s = capi.lib(), capi.stream()
        if self.bf16:   # one launch: prep of every step, the Split2d convs (f32) and the coupling nets' convs (bf16)
            rp = st["step_repack_dev"] if want_plain else st["step_repack_frag_dev"]
            cut = st["refresh_cut"]
            if (may_defer and self.defer_refresh and cut is not None and not want_inverse and trainable and self.aux_stream is not None and
                    self.aux is not None and self.level_wgrad_aux):
                pA, rA, _ = cut
                cur = torch.cuda.current_stream()
                fork = torch.cuda.Event()
                fork.record(cur)
                self.aux_stream.wait_event(fork)
                with torch.cuda.stream(self.aux_stream):
                    check(lib.nfk_prep_repack(ptr(st["prep_dev"]) + C.sizeof(PrepDesc) * pA, st["n_prep"] - pA, 0,
                                              ptr(st["repack_dev"]), st["n_repack"], ptr(rp) + C.sizeof(RepackDesc) * rA,
                                              st["n_step_repack"] - rA, capi.stream()), "nfk_prep_repack (deferred part)")
                    st["refresh_event"] = torch.cuda.Event()
                    st["refresh_event"].record(self.aux_stream)
                check(lib.nfk_prep_repack(ptr(st["prep_dev"]), pA, 0, None, 0, ptr(rp), rA, s), "nfk_prep_repack (first level)")
            else:
                check(lib.nfk_prep_repack(ptr(st["prep_dev"]), st["n_prep"], 1 if want_inverse else 0, ptr(st["repack_dev"]),
                                          st["n_repack"], ptr(rp), st["n_step_repack"], s), "nfk_prep_repack")
            st["has_plain"] = want_plain
        else:
            check(lib.nfk_prep(ptr(st["prep_dev"]), st["n_prep"], 1 if want_inverse else 0, s), "nfk_prep")
            check(lib.nfk_repack_batch(ptr(st["repack_dev"]), st["n_repack"], s), "nfk_repack_batch")
            check(lib.nfk_repack_batch(ptr(st["step_repack_dev"]), st["n_step_repack"], s), "nfk_repack_batch(step convs)")
            st["has_plain"] = True
        st["version"], st["has_inverse"] = ver, bool(want_inverse)

    # ------------------------------------------------------------------ public entry points
    def encode(self, x, y_onehot, logdet, noise=None, keep=None, hints_only=False):
        """-> (list of every layer's output, logdet).  `keep` (optional, honoured by passes that record no gradient): the
        layer indices whose outputs the caller will read (the KD hint layers of pl_module.py:262-282; negative indices
        allowed; the last layer is always kept) -- the other entries of the list are None and never reach HBM where a
        whole level runs as one launch."""
        if y_onehot is not None and not self.is_1d:
            # the reference concatenates a [B,classes] tensor to a 4-D one here (models/flows.py:153)
            raise RuntimeError("Tensors must have same number of dimensions: got 4 and 2")
        B = x.shape[0]
        if not torch.is_tensor(logdet):
            logdet = torch.full((B,), float(logdet), device=x.device, dtype=torch.float32)
        if keep is not None:
            n_layers = len(self.layers)
            keep = frozenset(int(k) % n_layers for k in keep) | {n_layers - 1}
        # (inside autograd.Function.forward the grad mode is always off: whether this pass records gradients is read here)
        # hints_only (with `keep`, passes that record no gradient): the caller reads the kept layer outputs and nothing else
        # -- the frozen teacher of the distillation step (pl_module.py:216-225 discards its bits/dim).  Split2d then only
        # passes z1 on (no prior convolution, no log-density) and the log-det partials are never summed; the returned logdet
        # is meaningless.
        res = _EncodeFn.apply(self, x, logdet, noise, y_onehot, (keep, torch.is_grad_enabled(), bool(hints_only and keep is not None)),
                              *self.params())
        return list(res[:-1]), res[-1]

    def decode(self, z, y_onehot, temperature, eps=None):
        """Reverse pass.  `eps`: {layer index: N(0,1) tensor [B,C/2,H,W]} for Split2d.  It carries gradients (the
        perceptual branch of the distillation loss, pl_module.py:229-243) when autograd is recording and a parameter or `z`
        requires them: wrt every FlowStep parameter and `z`.  Split2d's draw is torch.normal(mean, std) upstream
        (models/layers.py:26-29), which is not differentiable: its prior conv gets no gradient from a reverse pass."""
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.params())):
            if self.is_1d:
                return list(_DecodeFn.apply(self, z, y_onehot, *self.params()))
            return list(_Decode2dFn.apply(self, z, temperature, eps, *self.params()))
        with torch.no_grad():
            return self._decode_nograd(z, y_onehot, temperature, eps)

    def _decode_nograd(self, z, y_onehot, temperature, eps=None):
        st = self._ensure_for_decode(z)
        self._refresh(st, True)
        lib, s = capi.lib(), capi.stream()
        B = z.shape[0]
        outs = []
        if self.is_1d:
            cur = z.contiguous()
            for r in reversed(st["recs"]):
                cur = r.tab.inverse(cur, y_onehot)
                outs.append(cur)
            return outs
        cur = _nhwc(z)
        scratch = _Scratch(st, z.device)
        recs = st["recs"]
        for n, r in enumerate(reversed(recs)):
            last = n == len(recs) - 1
            if r.kind == "step":
                M_ = B * r.H * r.W
                out = torch.empty((B, r.H, r.W, r.C), device=z.device, dtype=torch.float32)
                zb, h1, h2, ws = scratch.get(M_ * r.C, M_ * r.hid, M_ * r.hid, lib.nfk_step2d_fwd_ws_floats(C.byref(r.cs)))
                check(lib.nfk_step2d_inv(C.byref(r.cs), ptr(cur), r.C, None, ptr(zb), ptr(h1), ptr(h2), None, ptr(ws),
                                         ptr(out), None, s), "nfk_step2d_inv")
                cur = out
                outs.append(out.permute(0, 3, 1, 2))
            elif r.kind == "split":
                out = torch.empty((B, r.H, r.W, r.C), device=z.device, dtype=torch.float32)
                e = None
                T = 0.0 if temperature is None else float(temperature)
                if eps is not None and r.idx in eps:
                    e = _nhwc(eps[r.idx].to(torch.float32))
                elif T != 0.0:
                    e = torch.randn((B, r.H, r.W, r.C // 2), device=z.device, dtype=torch.float32)
                (ws,) = scratch.get(lib.nfk_split2d_fwd_ws_floats(C.byref(r.cs)))
                check(lib.nfk_split2d_inv(C.byref(r.cs), ptr(cur), ptr(e), T, ptr(out), ptr(ws), s), "nfk_split2d_inv")
                cur = out
                outs.append(out.permute(0, 3, 1, 2))
            else:  # unsqueeze: r holds the UNsqueezed dims
                if last:
                    out = torch.empty((B, r.C, r.H, r.W), device=z.device, dtype=torch.float32)
                    check(lib.nfk_unsqueeze(ptr(cur), ptr(out), B, r.C, r.H, r.W, 1, s), "nfk_unsqueeze")
                    outs.append(out)
                else:
                    out = torch.empty((B, r.H, r.W, r.C), device=z.device, dtype=torch.float32)
                    check(lib.nfk_unsqueeze(ptr(cur), ptr(out), B, r.C, r.H, r.W, 0, s), "nfk_unsqueeze")
                    outs.append(out.permute(0, 3, 1, 2))
                cur = out
        return outs

    def decode_single(self, z, y_onehot, temperature, eps=None):
        return self.decode(z, y_onehot, temperature, eps)[-1]

    def _ensure_for_decode(self, z):
        """The engine is keyed on the INPUT shape of the first layer; derive it from the latent."""
        M = self.M
        if self.is_1d:
            return self._ensure(z)
        Cc, H, W = z.shape[1:]
        for l in reversed(self.layers):
            if isinstance(l, M.SqueezeLayer):
                Cc, H, W = Cc // 4, H * 2, W * 2
            elif isinstance(l, M.Split2d):
                Cc = Cc * 2
        capi.require_cuda(z, "z")
        key = (z.shape[0], (Cc, H, W), z.device)
        if self.state is None or self.key != key or self.state["sig"] != self._signature():
            self._build(z.shape[0], (Cc, H, W), z.device)
            self.key = key
        return self.state


class _Scratch:
    """Re-usable scratch tensors for passes that keep nothing (teacher forward, reverse pass)."""

    def __init__(self, st, device):
        self.device = device
        self.bufs = []

    def get(self, *sizes):
        out = []
        for i, n in enumerate(sizes):
            n = max(int(n), 4)
            if i >= len(self.bufs):
                self.bufs.append(torch.empty(n, device=self.device, dtype=torch.float32))
            elif self.bufs[i].numel() < n:
                self.bufs[i] = torch.empty(n, device=self.device, dtype=torch.float32)
            out.append(self.bufs[i])
        return out


def _cached_grad_desc(st, tag, pg, dev):
    """Device copy of the nfk_prep_grad descriptors, uploaded only when a pointer in them changed (the workspaces and the
    flat gradient arena are persistent, so after the first step this is a lookup: no host-to-device copy on the hot path,
    none inside a hipGraph capture)."""
    sig = tuple((d.part, d.nparts, d.g_an_bias, d.g_an_logs, d.g_lower, d.g_upper, d.g_log_s, d.g_weight) for d in pg)
    key = "grad_desc_" + tag
    cached = st.get(key)
    if cached is None or cached[0] != sig:
        cached = st[key] = (sig, capi.struct_array_to_device(pg, dev))
    return cached[1]


class _DecodeFn(torch.autograd.Function):
    """The reverse pass of a 1-D FlowNet (models/kd_flows.py:55-73) with gradients, as one autograd node: parameter
    gradients are accumulated straight into `.grad` (as _EncodeFn does); returns every intermediate tensor."""

    @staticmethod
    def forward(ctx, eng, z, y_onehot, *params):
        st = eng._ensure_for_decode(z)
        eng._refresh(st, True)
        ctx.set_materialize_grads(False)
        cur = z.detach().to(torch.float32).contiguous()
        outs, saved = [], []
        for r in reversed(st["recs"]):
            r.tab.part_mix_inv = None
            cur, sv = r.tab.inverse(cur, y_onehot, need=True)
            outs.append(cur)
            saved.append(sv)
        ctx.eng, ctx.saved, ctx.y_onehot, ctx.st = eng, saved, y_onehot, st
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        eng = ctx.eng
        st = ctx.st   # the state the forward ran with (another call may have rebuilt eng.state for another batch size)
        lib, s = capi.lib(), capi.stream()
        recs = st["recs"]
        dev = ctx.saved[0][0].device
        B = ctx.saved[0][0].shape[0]
        for p in eng.params():
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        want_gy = ctx.needs_input_grad[2]
        g, g_y = None, None
        n_steps = len(recs)
        for n in reversed(range(n_steps)):   # decode order n handled recs[n_steps - 1 - n]
            r = recs[n_steps - 1 - n]
            go = g_outs[n]
            if go is not None:
                go = go.to(torch.float32).contiguous()
                g = go if g is None else g + go
            if g is None:
                continue
            g, gc = r.tab.inverse_backward(ctx.saved[n], g, want_gy)
            if gc is not None:
                g_y = gc if g_y is None else g_y + gc
        pg = []
        for r in recs:
            d = PrepGrad()
            if r.tab.part_mix_inv is not None:
                if r.C > 74:
                    raise RuntimeError("reverse-pass gradients support at most 74 features per step")
                d.part, d.nparts = r.tab.part_mix_inv.data_ptr(), r.tab.nblk
                an, ic = r.mod.actnorm, getattr(r.mod, "invconv", None)
                d.g_an_bias, d.g_an_logs = ptr(an.bias.grad), ptr(an.logs.grad)
                if ic is None:
                    pass   # fixed permutation (MADE block): no parameters behind the mix
                elif ic.LU_decomposed:
                    d.g_lower, d.g_upper, d.g_log_s = ptr(ic.lower.grad), ptr(ic.upper.grad), ptr(ic.log_s.grad)
                else:
                    d.g_weight = ptr(ic.weight.grad)
            pg.append(d)
        if any(d.nparts > 0 for d in pg):
            grad_dev = _cached_grad_desc(st, "inv", pg, dev)
            check(lib.nfk_prep_inv_bwd(ptr(st["prep_dev"]), ptr(grad_dev), st["n_prep"], None, B, s), "nfk_prep_inv_bwd")
        gz = g if ctx.needs_input_grad[1] else None
        return (None, gz, g_y if want_gy else None) + (None,) * len(eng.params())


class _Decode2dFn(torch.autograd.Function):
    """The reverse pass of an image FlowNet (models/kd_flows.py:55-73) with gradients, as one autograd node: every FlowStep
    through nfk_step2d_inv / nfk_step2d_inv_bwd; Split2d concatenates a draw that carries no gradient (torch.normal
    upstream), so its backward is the first half of the incoming gradient; unsqueeze's backward is squeeze."""

    @staticmethod
    def forward(ctx, eng, z, temperature, eps, *params):
        st = eng._ensure_for_decode(z)
        eng._refresh(st, True)
        ctx.set_materialize_grads(False)
        lib, s = capi.lib(), capi.stream()
        B, dev = z.shape[0], z.device
        cur = _nhwc(z.detach().to(torch.float32))
        scratch = _Scratch(st, dev)
        recs = st["recs"]
        outs, saved = [], []
        for n, r in enumerate(reversed(recs)):
            last = n == len(recs) - 1
            if r.kind == "step":
                M_ = B * r.H * r.W
                out = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                o1 = _a4(M_ * r.C)
                o2 = o1 + _a4(M_ * r.hid)
                o3 = o2 + _a4(M_ * r.hid)
                arena = torch.empty(o3 + _a4(M_ * r.Cout), device=dev, dtype=torch.float32)
                zb, h1, h2, hs = arena[:o1], arena[o1:o2], arena[o2:o3], arena[o3:]
                (ws,) = scratch.get(lib.nfk_step2d_fwd_ws_floats(C.byref(r.cs)))
                check(lib.nfk_step2d_inv(C.byref(r.cs), ptr(cur), r.C, None, ptr(zb), ptr(h1), ptr(h2), ptr(hs), ptr(ws),
                                         ptr(out), None, s), "nfk_step2d_inv")
                saved.append((cur, zb, h1, h2, hs, arena))
                cur = out
                outs.append(out.permute(0, 3, 1, 2))
            elif r.kind == "split":
                out = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                e = None
                T = 0.0 if temperature is None else float(temperature)
                if eps is not None and r.idx in eps:
                    e = _nhwc(eps[r.idx].to(torch.float32))
                elif T != 0.0:
                    e = torch.randn((B, r.H, r.W, r.C // 2), device=dev, dtype=torch.float32)
                (ws,) = scratch.get(lib.nfk_split2d_fwd_ws_floats(C.byref(r.cs)))
                check(lib.nfk_split2d_inv(C.byref(r.cs), ptr(cur), ptr(e), T, ptr(out), ptr(ws), s), "nfk_split2d_inv")
                saved.append(None)
                cur = out
                outs.append(out.permute(0, 3, 1, 2))
            else:  # unsqueeze: r holds the UNsqueezed dims
                if last:
                    out = torch.empty((B, r.C, r.H, r.W), device=dev, dtype=torch.float32)
                    check(lib.nfk_unsqueeze(ptr(cur), ptr(out), B, r.C, r.H, r.W, 1, s), "nfk_unsqueeze")
                    outs.append(out)
                else:
                    out = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                    check(lib.nfk_unsqueeze(ptr(cur), ptr(out), B, r.C, r.H, r.W, 0, s), "nfk_unsqueeze")
                    outs.append(out.permute(0, 3, 1, 2))
                saved.append(None)
                cur = out
        ctx.eng, ctx.st, ctx.saved, ctx.B = eng, st, saved, B
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        eng, st, B = ctx.eng, ctx.st, ctx.B
        lib, s = capi.lib(), capi.stream()
        recs = st["recs"]
        dec = list(reversed(recs))           # decode order: output n came from dec[n]
        dev = st["prep_dev"].device
        for p in eng.params():
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        scratch = _Scratch(st, dev)
        part_off, total = {}, 0
        for r in recs:
            if r.kind == "step":
                if r.C > 74:
                    raise RuntimeError("reverse-pass gradients support at most 74 channels per step")
                nb = lib.nfk_chanmix_blocks(B * r.H * r.W, r.C)
                part_off[r.idx] = (total, nb)
                total += _a4(nb * (r.C * r.C + r.C))
        parts = torch.empty(max(total, 4), device=dev, dtype=torch.float32)
        used = set()
        g = None   # dL/d(output of dec[n]) in that output's memory layout (NHWC; NCHW for the last layer)
        for n in reversed(range(len(dec))):
            r, go = dec[n], g_outs[n]
            last = n == len(dec) - 1
            if go is not None:
                go = go.to(torch.float32)
                go = go.contiguous() if (last and r.kind == "squeeze") else _nhwc(go)
                g = go if g is None else g + go
            if g is None:
                continue
            if r.kind == "step":
                y, zb, h1, h2, hs, _ = ctx.saved[n]
                blk = r.mod.block
                gs = Step2dGrads()
                gs.g_w1, gs.g_b1, gs.g_l1 = ptr(blk[0].conv.weight.grad), ptr(blk[0].actnorm.bias.grad), ptr(blk[0].actnorm.logs.grad)
                gs.g_w2, gs.g_b2, gs.g_l2 = ptr(blk[2].conv.weight.grad), ptr(blk[2].actnorm.bias.grad), ptr(blk[2].actnorm.logs.grad)
                gs.g_w3, gs.g_bias3, gs.g_l3 = ptr(blk[4].conv.weight.grad), ptr(blk[4].conv.bias.grad), ptr(blk[4].logs.grad)
                off, nb = part_off[r.idx]
                gs.mix_part = parts.data_ptr() + 4 * off
                gy = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                (ws,) = scratch.get(lib.nfk_step2d_bwd_ws_floats(C.byref(r.cs)))
                check(lib.nfk_step2d_inv_bwd(C.byref(r.cs), C.byref(gs), ptr(y), r.C, ptr(zb), ptr(h1), ptr(h2), ptr(hs),
                                             ptr(g), None, ptr(gy), ptr(ws), s), "nfk_step2d_inv_bwd")
                used.add(r.idx)
                g = gy
            elif r.kind == "split":
                g = g[..., : r.C // 2].contiguous()   # z2 is a draw: no gradient reaches the prior conv or z1 through it
            else:  # unsqueeze: its input was the squeezed NHWC tensor
                gi = torch.empty((B, r.H // 2, r.W // 2, 4 * r.C), device=dev, dtype=torch.float32)
                check(lib.nfk_squeeze(ptr(g), r.C, None, ptr(gi), B, r.C, r.H, r.W, 1 if last else 0, s), "nfk_squeeze")
                g = gi
        pg = []
        for r in recs:
            d = PrepGrad()
            if r.kind == "step" and r.idx in used:
                off, nb = part_off[r.idx]
                d.part, d.nparts = parts.data_ptr() + 4 * off, nb
                an = r.mod.actnorm
                d.g_an_bias, d.g_an_logs = ptr(an.bias.grad), ptr(an.logs.grad)
                if r.mod.flow_permutation_type == "invconv":
                    ic = r.mod.invconv
                    if ic.LU_decomposed:
                        d.g_lower, d.g_upper, d.g_log_s = ptr(ic.lower.grad), ptr(ic.upper.grad), ptr(ic.log_s.grad)
                    else:
                        d.g_weight = ptr(ic.weight.grad)
            if r.kind in ("step", "split"):
                pg.append(d)
        if any(d.nparts > 0 for d in pg):
            grad_dev = _cached_grad_desc(st, "inv2d", pg, dev)
            check(lib.nfk_prep_inv_bwd(ptr(st["prep_dev"]), ptr(grad_dev), st["n_prep"], None, B, s), "nfk_prep_inv_bwd")
        ctx.keep = (parts,)
        gz = None
        if ctx.needs_input_grad[1] and g is not None:
            gz = g.permute(0, 3, 1, 2)
        return (None, gz, None, None) + (None,) * len(eng.params())


class _EncodeFn(torch.autograd.Function):
    """All layers of a FlowNet, forward (models/kd_flows.py:42-53) and backward, as one autograd node."""

    @staticmethod
    def forward(ctx, eng, x, logdet, noise, y_onehot, keep_and_mode, *params):
        st = eng._ensure(x)
        eng._refresh(st, False, may_defer=True)
        lib, s = capi.lib(), capi.stream()
        B, dev = x.shape[0], x.device
        keep, grad_mode, hints_only = keep_and_mode
        need = grad_mode and any(ctx.needs_input_grad)   # under torch.no_grad() nothing is kept for a backward
        if need:
            keep = None   # a pass that will be differentiated keeps every layer's output
            hints_only = False
        ctx.set_materialize_grads(False)
        ctx.eng, ctx.need, ctx.B, ctx.st = eng, need, B, st
        ld = logdet.to(torch.float32).contiguous()
        outs, saved = [], []
        if eng.is_1d:
            cur = x.contiguous()
            for r in st["recs"]:
                r.tab.part_mix = None
                cur, ld, sv = r.tab.forward(cur, ld, y_onehot, need)
                outs.append(cur)
                saved.append(sv)
            ctx.saved, ctx.x_in = saved, x
            ctx.y_onehot = y_onehot
            return tuple(outs) + (ld,)

        scratch = _Scratch(st, dev)
        cur, cur_ld = None, None  # NHWC tensor of the running activation
        cur_stride = None         # row length of `cur` in floats when it is a channel slice of a wider tensor (hints_only)
        run, keep_alive = None, []
        lv_saved = {}   # {index of a level's last FlowStep: its strided arenas}: the backward of a whole level (nfk_level2d_bwd)
        skip_until = 0
        # log-det partials of the fused launches (a level's steps, the fused Split2d) are summed onto `ld` in ONE launch where
        # the running value is next needed -- normally at the end of the pass: same arithmetic in the same order as one
        # finishing launch per buffer (nfk_ld_finish_multi)
        pending = []

        def flush_ld(cur_ld):
            out_ld = cur_ld
            while pending:
                chunk = pending[:capi.LD_MAX_PARTS]
                del pending[:capi.LD_MAX_PARTS]
                arr = (capi.LdParts * len(chunk))()
                for k_, (buf_, ns_, tl_) in enumerate(chunk):
                    arr[k_].part, arr[k_].nsteps, arr[k_].tiles = buf_.data_ptr(), ns_, tl_
                    keep_alive.append(buf_)
                nxt = torch.empty_like(out_ld)
                check(lib.nfk_ld_finish_multi(ptr(out_ld), arr, len(chunk), B, ptr(nxt), s), "nfk_ld_finish_multi")
                out_ld = nxt
            return out_ld

        use_runs = eng.fused and eng.bf16 and (eng.fused_runs is True or (eng.fused_runs == "inference" and not need))
        for r in st["recs"]:
            if r.idx < skip_until:
                continue   # part of a run of steps that one launch has already walked
            if st["refresh_event"] is not None and r.idx >= st["refresh_cut"][2]:
                torch.cuda.current_stream().wait_event(st["refresh_event"])   # the deferred half of _refresh
                st["refresh_event"] = None
            if r.kind == "step" and use_runs and r.idx in st["runs"] and run is None:
                # ---- a whole level's FlowSteps in ONE launch (whole-image pixel tiles: z stays in LDS between the steps)
                info = st["runs"][r.idx]
                n = info["n"]
                if cur is None:
                    cur = _nhwc(x)
                tiles = lib.nfk_step2d_fused_tiles(C.byref(r.cs))
                M_ = B * r.H * r.W
                zc, hc = _a4(M_ * r.C), _a4(M_ * r.Cout)
                ldbuf = torch.empty(n * B * tiles, device=dev, dtype=torch.float32)
                fr = capi.FusedRun()
                fr.steps, fr.nsteps, fr.x, fr.ldx = info["dev"].data_ptr(), n, cur.data_ptr(), r.C
                fr.ldpart = ldbuf.data_ptr()
                if need:
                    kept = list(range(n))
                    zs_all = torch.empty(n * zc, device=dev, dtype=torch.float32)
                    h_all = torch.empty(2 * n * M_ * r.hid, device=dev, dtype=torch.bfloat16)
                    hs_all = torch.empty(n * hc, device=dev, dtype=torch.float32)
                    mbytes = lib.nfk_step2d_mask_bytes(C.byref(r.cs)) if r.cs.wg1 else 0
                    mk_all = torch.empty(n * mbytes, device=dev, dtype=torch.uint8) if mbytes else None
                    fr.zsave, fr.zsave_stride = zs_all.data_ptr(), zc
                    fr.h1save, fr.h2save, fr.h_stride = h_all.data_ptr(), h_all.data_ptr() + 2 * n * M_ * r.hid, M_ * r.hid
                    fr.hsave, fr.hsave_stride = hs_all.data_ptr(), hc
                    if mk_all is not None:
                        fr.masksave, fr.mask_stride = mk_all.data_ptr(), mbytes
                else:
                    kept = [k for k in range(n) if keep is None or (r.idx + k) in keep or k == n - 1]
                zo_all = torch.empty(len(kept) * zc, device=dev, dtype=torch.float32)
                fr.zout, fr.zout_stride = zo_all.data_ptr(), zc
                fr.keep = sum(1 << k for k in kept)
                check(lib.nfk_run2d_fwd_fused(C.byref(fr), C.byref(r.cs), s), "nfk_run2d_fwd_fused")
                pending.append((ldbuf, n, tiles))
                if need and mk_all is not None:   # what nfk_level2d_bwd addresses as base + k * stride
                    lv_saved[r.idx + n - 1] = {"first": r.idx, "n": n, "x0": cur, "zc": zc, "hc": hc, "zo_all": zo_all,
                                               "zs_all": zs_all, "h_all": h_all, "hs_all": hs_all, "mk_all": mk_all,
                                               "mbytes": mbytes, "mh": M_ * r.hid}
                slot = 0
                for k in range(n):
                    out = None
                    if k in kept:
                        out = zo_all[slot * zc: slot * zc + M_ * r.C].view(B, r.H, r.W, r.C)
                        slot += 1
                    if need:
                        h1 = h_all[k * M_ * r.hid: (k + 1) * M_ * r.hid]
                        h2 = h_all[(n + k) * M_ * r.hid: (n + k + 1) * M_ * r.hid]
                        masks = mk_all[k * mbytes: (k + 1) * mbytes] if mk_all is not None else None
                        saved.append((cur, zs_all[k * zc: k * zc + M_ * r.C], h1, h2, hs_all[k * hc: k * hc + M_ * r.Cout],
                                      None, masks))
                    else:
                        saved.append(None)
                    # (the run's last output always exists -- the next layer reads it -- but is only handed out if asked for)
                    handed = out is not None and (keep is None or (r.idx + k) in keep)
                    outs.append(out.permute(0, 3, 1, 2) if handed else None)
                    if out is not None:
                        cur = out
                skip_until = r.idx + n
                continue
            if r.kind == "squeeze":
                out = torch.empty((B, r.H // 2, r.W // 2, 4 * r.C), device=dev, dtype=torch.float32)
                if cur is None:
                    if x.is_contiguous():
                        check(lib.nfk_squeeze(ptr(x), 0, ptr(noise), ptr(out), B, r.C, r.H, r.W, 1, s), "nfk_squeeze")
                    else:
                        if noise is not None:
                            x.add_(noise)
                        xs = _nhwc(x)
                        check(lib.nfk_squeeze(ptr(xs), r.C, None, ptr(out), B, r.C, r.H, r.W, 0, s), "nfk_squeeze")
                else:
                    check(lib.nfk_squeeze(ptr(cur), cur_stride or r.C, None, ptr(out), B, r.C, r.H, r.W, 0, s), "nfk_squeeze")
                    cur_stride = None
                saved.append(None)
            elif r.kind == "step" and eng.fused and eng.bf16 and lib.nfk_step2d_fused_tiles(C.byref(r.cs)) > 0:
                # ---- fused path: one launch per step; the log-det is accumulated once per run of steps
                if cur is None:
                    cur = _nhwc(x)
                tiles = lib.nfk_step2d_fused_tiles(C.byref(r.cs))
                if run is None:
                    n_run = 0
                    for r2 in st["recs"][r.idx:]:
                        if r2.kind != "step" or (r2.C, r2.H, r2.W) != (r.C, r.H, r.W):
                            break
                        n_run += 1
                    run = {"buf": torch.empty(n_run * B * tiles, device=dev, dtype=torch.float32), "k": 0, "n": n_run,
                           "tiles": tiles, "lv": None}
                    lvi = st["levels"].get(r.idx)
                    if need and r.cs.wg1 and lvi is not None and lvi["n"] == n_run and lvi["bwd_ok"]:
                        # everything the steps of this level keep, in arenas addressed as base + k * stride (the layout of a
                        # level that runs as one launch): what the backward of the whole level reads (nfk_level2d_bwd)
                        Ml = B * r.H * r.W
                        zc_, hc_ = _a4(Ml * r.C), _a4(Ml * r.Cout)
                        mb_ = lib.nfk_step2d_mask_bytes(C.byref(r.cs))
                        run["lv"] = {"first": r.idx, "n": n_run, "x0": cur, "zc": zc_, "hc": hc_, "mbytes": mb_, "mh": Ml * r.hid,
                                     "zo_all": torch.empty(n_run * zc_, device=dev, dtype=torch.float32),
                                     "zs_all": torch.empty(n_run * zc_, device=dev, dtype=torch.float32),
                                     "h_all": torch.empty(2 * n_run * Ml * r.hid, device=dev, dtype=torch.bfloat16),
                                     "hs_all": torch.empty(n_run * hc_, device=dev, dtype=torch.float32),
                                     "mk_all": torch.empty(n_run * mb_, device=dev, dtype=torch.uint8)}
                M_ = B * r.H * r.W
                lvr = run["lv"]
                if lvr is not None:
                    k_ = run["k"]
                    out = lvr["zo_all"][k_ * lvr["zc"]: k_ * lvr["zc"] + M_ * r.C].view(B, r.H, r.W, r.C)
                else:
                    out = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                zb = h1 = h2 = hs = masks = None
                if need and lvr is not None:
                    k_, n_ = run["k"], run["n"]
                    zb = lvr["zs_all"][k_ * lvr["zc"]: k_ * lvr["zc"] + M_ * r.C]
                    h1 = lvr["h_all"][k_ * lvr["mh"]: (k_ + 1) * lvr["mh"]]
                    h2 = lvr["h_all"][(n_ + k_) * lvr["mh"]: (n_ + k_ + 1) * lvr["mh"]]
                    hs = lvr["hs_all"][k_ * lvr["hc"]: k_ * lvr["hc"] + M_ * r.Cout]
                    masks = lvr["mk_all"][k_ * lvr["mbytes"]: (k_ + 1) * lvr["mbytes"]]
                    saved.append((cur, zb, h1, h2, hs, None, masks))
                elif need:
                    # ReLU masks of h1 / h2 (one bit per element) for the fused step backward, behind the other saves
                    nmask = _a4((lib.nfk_step2d_mask_bytes(C.byref(r.cs)) + 3) // 4) if r.cs.wg1 else 0
                    arena = torch.empty(_a4(M_ * r.C) + 2 * _a4(M_ * r.hid) + _a4(M_ * r.Cout) + nmask, device=dev,
                                        dtype=torch.float32)
                    o1 = _a4(M_ * r.C)
                    o2 = o1 + _a4(M_ * r.hid)
                    o3 = o2 + _a4(M_ * r.hid)
                    o4 = o3 + _a4(M_ * r.Cout)
                    zb, h1, h2, hs = arena[:o1], arena[o1:o2], arena[o2:o3], arena[o3:o4]
                    masks = arena[o4:] if nmask else None
                    saved.append((cur, zb, h1, h2, hs, arena, masks))
                else:
                    saved.append(None)
                # hidden split of this launch: the engine's policy for a training / an inference forward
                ways = eng.fused_split_training if need else eng.fused_split_inference
                if ways == "frugal":   # beside another chain: only the grids that would fill fewer than 32 CUs
                    ways = 4 if (B * r.H * r.W + 63) // 64 < 32 else 0
                ways = int(ways)
                nws = lib.nfk_step2d_fused_ws_floats(C.byref(r.cs)) if ways else 0
                ws = None
                if nws:   # hidden-split step: partial conv3 slabs + (always) the mixed z for the coupling launch
                    if zb is None:
                        zb, ws = scratch.get(M_ * r.C, nws)
                    else:
                        (ws,) = scratch.get(nws)
                part = run["buf"].data_ptr() + 4 * run["k"] * B * tiles
                check(lib.nfk_step2d_fwd_fused(C.byref(r.cs), ptr(cur), r.C, ptr(zb), ptr(h1), ptr(h2), ptr(hs), ptr(out),
                                               part, ptr(ws), ptr(masks), ways, s), "nfk_step2d_fwd_fused")
                run["k"] += 1
                if run["k"] == run["n"]:
                    pending.append((run["buf"], run["n"], tiles))
                    if run["lv"] is not None:
                        lv_saved[r.idx] = run["lv"]
                    run = None
            elif r.kind == "step":
                if cur is None:
                    cur = _nhwc(x)
                M_ = B * r.H * r.W
                out = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                ld = flush_ld(ld)   # this launch reads the running log-det
                ld_out = torch.empty_like(ld)
                nws = lib.nfk_step2d_fwd_ws_floats(C.byref(r.cs))
                if need:
                    arena = torch.empty(_a4(M_ * r.C) + 2 * _a4(M_ * r.hid) + _a4(M_ * r.Cout), device=dev,
                                        dtype=torch.float32)
                    o1 = _a4(M_ * r.C)
                    o2 = o1 + _a4(M_ * r.hid)
                    o3 = o2 + _a4(M_ * r.hid)
                    zb, h1, h2, hs = arena[:o1], arena[o1:o2], arena[o2:o3], arena[o3:]
                    (ws,) = scratch.get(nws)
                    saved.append((cur, zb, h1, h2, hs, arena, None))
                else:
                    zb, h1, h2, ws = scratch.get(M_ * r.C, M_ * r.hid, M_ * r.hid, nws)
                    hs = None
                    saved.append(None)
                check(lib.nfk_step2d_fwd(C.byref(r.cs), ptr(cur), r.C, ptr(ld), ptr(zb), ptr(h1), ptr(h2), ptr(hs),
                                         ptr(ws), ptr(out), ptr(ld_out), s), "nfk_step2d_fwd")
                ld = ld_out
            elif hints_only and r.idx not in keep and r.idx + 1 < len(st["recs"]) and st["recs"][r.idx + 1].kind == "squeeze":
                # hints-only pass: Split2d passes z1 = the first half of the channels on; the squeeze that follows reads it in
                # place, with the full rows' stride
                if cur is None:
                    cur = _nhwc(x)
                out = cur[..., : r.C // 2]
                cur_stride = r.C
                saved.append(None)
            else:  # split
                if cur is None:
                    cur = _nhwc(x)
                M_ = B * r.H * r.W
                out = torch.empty((B, r.H, r.W, r.C // 2), device=dev, dtype=torch.float32)
                hs = torch.empty(M_ * r.C, device=dev, dtype=torch.float32) if need else None
                sp_tiles = lib.nfk_split2d_fwd_tiles(C.byref(r.cs))
                if sp_tiles > 0:   # one fused launch that leaves per-tile partials: finished with the others
                    lbuf = torch.empty(max(B * sp_tiles, 4), device=dev, dtype=torch.float32)
                    check(lib.nfk_split2d_fwd(C.byref(r.cs), ptr(cur), None, ptr(out), ptr(hs), ptr(lbuf), None, s),
                          "nfk_split2d_fwd")
                    pending.append((lbuf, 1, sp_tiles))
                else:
                    ld = flush_ld(ld)
                    ld_out = torch.empty_like(ld)
                    (ws,) = scratch.get(lib.nfk_split2d_fwd_ws_floats(C.byref(r.cs)))
                    check(lib.nfk_split2d_fwd(C.byref(r.cs), ptr(cur), ptr(ld), ptr(out), ptr(hs), ptr(ws), ptr(ld_out), s),
                          "nfk_split2d_fwd")
                    ld = ld_out
                saved.append((cur, hs) if need else None)
            cur = out
            outs.append(out.permute(0, 3, 1, 2) if (keep is None or r.idx in keep) else None)
            if r is st["recs"][0] and eng.first_layer_hook is not None:
                eng.first_layer_hook()  # the model input has been consumed (and dequantised in place)
        if not hints_only:
            ld = flush_ld(ld)
        ctx.saved = saved
        ctx.levels = lv_saved
        ctx.x_contig = x.is_contiguous()
        return tuple(outs) + (ld,)

    @staticmethod
    def backward(ctx, *grads):
        eng, B = ctx.eng, ctx.B
        st = ctx.st   # the state the forward ran with (another call may have rebuilt eng.state for another batch size)
        lib, s = capi.lib(), capi.stream()
        recs = st["recs"]
        g_outs, g_ld = grads[:-1], grads[-1]
        dev = st["prep_buf"].device
        if g_ld is None:
            g_ld = torch.zeros(B, device=dev, dtype=torch.float32)
        g_ld = g_ld.to(torch.float32).contiguous()
        n_in = 6 + len(eng.params())
        # make sure every parameter has a grad buffer the kernels can accumulate into
        for p in eng.params():
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        if eng.is_1d:
            g, g_y = None, None
            want_gy = ctx.needs_input_grad[4]
            for r, sv, go in zip(reversed(recs), reversed(ctx.saved), reversed(g_outs)):
                if go is not None:
                    g = go.to(torch.float32).contiguous() if g is None else g + go
                if g is None:
                    continue
                g, gc = r.tab.backward(sv, g, g_ld, want_gy)
                if gc is not None:
                    g_y = gc if g_y is None else g_y + gc
            pg = []
            for r in recs:
                d = PrepGrad()
                if r.tab.part_mix is not None:
                    d.part, d.nparts = r.tab.part_mix.data_ptr(), r.tab.nblk
                    an, ic = r.mod.actnorm, getattr(r.mod, "invconv", None)
                    d.g_an_bias, d.g_an_logs = ptr(an.bias.grad), ptr(an.logs.grad)
                    if ic is None:
                        pass   # fixed permutation (MADE block): no parameters behind the mix
                    elif ic.LU_decomposed:
                        d.g_lower, d.g_upper, d.g_log_s = ptr(ic.lower.grad), ptr(ic.upper.grad), ptr(ic.log_s.grad)
                    else:
                        d.g_weight = ptr(ic.weight.grad)
                pg.append(d)
            if all(d.nparts > 0 for d in pg):
                grad_dev = _cached_grad_desc(st, "fwd", pg, dev)
                check(lib.nfk_prep_bwd(ptr(st["prep_dev"]), ptr(grad_dev), st["n_prep"], ptr(g_ld), B, st["any_dense"], s),
                      "nfk_prep_bwd")
            elif any(d.nparts > 0 for d in pg):
                raise RuntimeError("partial backward through a 1-D flow is not supported")
            gx = g if ctx.needs_input_grad[1] else None
            return (None, gx, g_ld if ctx.needs_input_grad[2] else None, None, g_y) + (None,) * (n_in - 5)

        scratch = _Scratch(st, dev)
        # per-step partial (gWf, gcf) arena for nfk_prep_bwd
        part_off, total = {}, 0
        levels_ok = (eng.level_backward and eng.bf16 and eng.fused and eng.bwd_mode in (capi.BWD_DEFAULT, capi.BWD_FUSED)
                     and bool(getattr(ctx, "levels", None)))
        for r in recs:
            if r.kind == "step":
                nb = lib.nfk_chanmix_blocks(B * r.H * r.W, r.C)
                part_off[r.idx] = (total, nb)
                if levels_ok:
                    nb = max(nb, lib.nfk_level_bwd_parts(C.byref(r.cs)))
                total += _a4(nb * (r.C * r.C + r.C))
        parts = st.get("parts")
        if parts is None or parts.numel() < max(total, 4):
            parts = st["parts"] = torch.empty(max(total, 4), device=dev, dtype=torch.float32)
        nparts_used = {}   # {layer index: partial slabs its backward left in `parts`} where that is not part_off's count

        # With an auxiliary stream (training.enable_concurrency) and no per-layer gradient hook, the weight-gradient
        # GEMMs of ALL steps form one chain on that stream, fed by per-step gradient buffers, and are joined once at
        # the end: the dgrad chain on this stream never waits for them.
        # (bf16: the dual / fused launches keep everything on one stream unless defer_fused_wgrad asks for the separate
        #  weight-gradient chain -- bit-identical, but measured slower inside a replayed hipGraph: 24 more cross-stream edges)
        deferred = bool(eng.aux) and eng.grad_hook is None and (not eng.bf16 or eng.defer_fused_wgrad)
        ga_off, ga_total, wslab_max = {}, 0, 4
        if deferred:
            for r in recs:
                if r.kind == "step":
                    ga_off[r.idx] = ga_total
                    ga_total += _a4(lib.nfk_step2d_bwd_ga_floats(C.byref(r.cs)))
                    wslab_max = max(wslab_max, lib.nfk_step2d_bwd_wslab_floats(C.byref(r.cs)))
            ga_all, wslab = st.get("ga_all"), st.get("wslab")
            if ga_all is None or ga_all.numel() < max(ga_total, 4):
                ga_all = st["ga_all"] = torch.empty(max(ga_total, 4), device=dev, dtype=torch.float32)
            if wslab is None or wslab.numel() < wslab_max:
                wslab = st["wslab"] = torch.empty(wslab_max, device=dev, dtype=torch.float32)

        # Steps are walked as a chain (nfk_step2d_bwd_chain): the launch that ends a step's backward also runs the
        # coupling backward of the step that follows on the same level, into that step's workspace -- hence two
        # workspaces used alternately.
        ws_max = max([lib.nfk_step2d_bwd_ws_floats(C.byref(r.cs)) for r in recs if r.kind == "step"] + [4])
        ws_pair = [torch.empty(ws_max, device=dev, dtype=torch.float32) for _ in range(2)] if eng.chain_backward else None
        order = list(zip(reversed(recs), reversed(ctx.saved), reversed(g_outs)))
        ext_of = {}
        # (a per-layer gradient hook wants every layer's gradients final when its call returns: no owed reductions then)
        pend = capi.Step2dPending() if (eng.merge_wgrad_tail and eng.bf16 and eng.grad_hook is None and ws_pair is not None) else None

        def _ext(i):
            if i not in ext_of:
                go_i = order[i][2]
                ext_of[i] = _nhwc(go_i.to(torch.float32)) if go_i is not None else None
            return ext_of[i]

        def level_backward(i, lvl, g):
            """order[i] is the LAST FlowStep of a level whose forward kept its tensors in strided arenas: the whole level
            through nfk_level2d_bwd (cut where a step's output carries an external gradient: such a step starts a new
            walk).  Returns the gradient wrt the level's input."""
            n, first = lvl["n"], lvl["first"]
            info = st["levels"][first]
            r0 = recs[first]
            M_ = B * r0.H * r0.W
            npart = lib.nfk_level_bwd_parts(C.byref(r0.cs))
            # per-step gradient pointers (persistent buffers: uploaded once per set of pointers)
            lg = []
            for q in recs[first:first + n]:
                blk = q.mod.block
                d = capi.LevelGrads()
                d.g_w1, d.g_b1, d.g_l1 = ptr(blk[0].conv.weight.grad), ptr(blk[0].actnorm.bias.grad), ptr(blk[0].actnorm.logs.grad)
                d.g_w2, d.g_b2, d.g_l2 = ptr(blk[2].conv.weight.grad), ptr(blk[2].actnorm.bias.grad), ptr(blk[2].actnorm.logs.grad)
                d.g_w3, d.g_bias3, d.g_l3 = ptr(blk[4].conv.weight.grad), ptr(blk[4].conv.bias.grad), ptr(blk[4].logs.grad)
                d.mix_part = parts.data_ptr() + 4 * part_off[q.idx][0]
                lg.append(d)
                nparts_used[q.idx] = npart
            sig = tuple((d.g_w1, d.g_b1, d.g_l1, d.g_w2, d.g_b2, d.g_l2, d.g_w3, d.g_bias3, d.g_l3, d.mix_part) for d in lg)
            cache = st.setdefault("level_grads", {})
            if first not in cache or cache[first][0] != sig:
                cache[first] = (sig, capi.struct_array_to_device(lg, dev))
            wsz = lib.nfk_level_bwd_ws_floats(C.byref(r0.cs), n)
            wsc = st.setdefault("level_ws", {})
            if first not in wsc or wsc[first].numel() < wsz:
                wsc[first] = torch.empty(wsz, device=dev, dtype=torch.float32)
            lv = capi.LevelBwd()
            lv.steps, lv.grads, lv.nsteps = info["dev"].data_ptr(), cache[first][1].data_ptr(), n
            lv.x0, lv.ldx0 = lvl["x0"].data_ptr(), r0.C
            lv.zout, lv.zout_stride = lvl["zo_all"].data_ptr(), lvl["zc"]
            lv.zsave, lv.zsave_stride = lvl["zs_all"].data_ptr(), lvl["zc"]
            lv.hsave, lv.hsave_stride = lvl["hs_all"].data_ptr(), lvl["hc"]
            lv.h1save, lv.h2save, lv.h_stride = lvl["h_all"].data_ptr(), lvl["h_all"].data_ptr() + 2 * n * lvl["mh"], lvl["mh"]
            lv.masks, lv.mask_stride = lvl["mk_all"].data_ptr(), lvl["mbytes"]
            lv.g_ld, lv.ws = ptr(g_ld), wsc[first].data_ptr()
            lv.multi = 1 if eng.level_multi else 0
            lv.aux = eng.aux if (eng.level_wgrad_aux and eng.grad_hook is None) else None
            lv.wgrad_chunk = int(eng.level_wgrad_chunk) if lv.aux else 0
            # with the auxiliary stream the level's parameter-side chain (nfk_prep_bwd of its steps) follows its reductions
            # THERE, under the next level's launches, instead of waiting for the end of the backward (where only the first
            # level's is left: the narrow steps, whose LU chains are the short ones)
            lpg_dev = None
            if lv.aux:
                lpg = [_prep_grad(q, npart) for q in recs[first:first + n]]
                psig = tuple((d.part, d.nparts, d.g_an_bias, d.g_lower, d.g_weight) for d in lpg)
                pcache = st.setdefault("level_prep_grads", {})
                if first not in pcache or pcache[first][0] != psig:
                    pcache[first] = (psig, capi.struct_array_to_device(lpg, dev))
                lpg_dev = pcache[first][1]
                prep_done.update(q.idx for q in recs[first:first + n])
                if first == min(st["levels"]) and not deferred:
                    # the level walked LAST: its weight gradients (the longest launch of the backward) have nothing but the
                    # optimiser behind them, and its parameter-side chain does not read them -- that chain goes out on the
                    # caller's stream, beside them, instead of behind them
                    last_prep.append((st["prep_dev"].data_ptr() + C.sizeof(capi.PrepDesc) * r0.prep_i, lpg_dev, n))
                    lpg_dev = None
            k_hi = n - 1
            while k_hi >= 0:
                ext = _ext(i + (n - 1 - k_hi))
                k_lo = k_hi
                while k_lo > 0 and _ext(i + (n - 1 - (k_lo - 1))) is None:
                    k_lo -= 1
                g_main, g_e = (g, ext) if g is not None else (ext, None)
                gx = torch.empty((B, r0.H, r0.W, r0.C), device=dev, dtype=torch.float32)
                lv.g_out, lv.g_ext, lv.g_x = ptr(g_main), ptr(g_e), ptr(gx)
                lv.k_lo, lv.k_hi = k_lo, k_hi
                if lpg_dev is not None:
                    lv.prep_descs = st["prep_dev"].data_ptr() + C.sizeof(capi.PrepDesc) * (r0.prep_i + k_lo)
                    lv.prep_grads = lpg_dev.data_ptr() + C.sizeof(PrepGrad) * k_lo
                    lv.prep_n, lv.prep_dense = k_hi - k_lo + 1, int(st["any_dense"])
                check(lib.nfk_level2d_bwd(C.byref(lv), info["host"], s), "nfk_level2d_bwd")
                if lv.aux:
                    aux_used.append(True)
                keep_alive.append((g_main, g_e))
                g = gx
                k_hi = k_lo - 1
            if eng.grad_hook is not None:
                eng.grad_hook(first)   # every layer of the level at once: its gradients are final behind the level's last launch
            return g

        def _prep_grad(q, nparts):
            d = PrepGrad()
            off, _ = part_off[q.idx]
            d.part, d.nparts = parts.data_ptr() + 4 * off, nparts
            an = q.mod.actnorm
            d.g_an_bias, d.g_an_logs = ptr(an.bias.grad), ptr(an.logs.grad)
            if q.mod.flow_permutation_type == "invconv":
                ic = q.mod.invconv
                if ic.LU_decomposed:
                    d.g_lower, d.g_upper, d.g_log_s = ptr(ic.lower.grad), ptr(ic.upper.grad), ptr(ic.log_s.grad)
                else:
                    d.g_weight = ptr(ic.weight.grad)
            return d

        aux_used, keep_alive, prep_done, last_prep = [], [], set(), []
        g = None  # grad wrt the current layer's OUTPUT coming from downstream (NHWC)
        rows_in, n_step = 0, 0   # partial rows the previous call left for this step (0: none); steps processed
        skip_to = 0
        for i, (r, sv, go) in enumerate(order):
            if i < skip_to:
                continue   # part of a level that one call has walked
            ext = _ext(i)
            if g is None and ext is None:
                continue  # nothing flows into this layer's output
            if r.kind == "step" and levels_ok and r.idx in ctx.levels and st["levels"][ctx.levels[r.idx]["first"]]["bwd_ok"]:
                if pend is not None:   # what a per-step call before this level still owes
                    check(lib.nfk_step2d_pending_flush(C.byref(pend), s), "nfk_step2d_pending_flush")
                lvl = ctx.levels[r.idx]
                g = level_backward(i, lvl, g)
                skip_to = i + lvl["n"]
                rows_in = 0
            elif r.kind == "step" and ws_pair is not None:
                x_in, zb, h1, h2, hs, _, masks = sv
                blk = r.mod.block
                gs = Step2dGrads()
                gs.relu_masks = ptr(masks)
                gs.bwd_mode, gs.aux = eng.bwd_mode, eng.aux
                gs.g_w1, gs.g_b1, gs.g_l1 = ptr(blk[0].conv.weight.grad), ptr(blk[0].actnorm.bias.grad), ptr(blk[0].actnorm.logs.grad)
                gs.g_w2, gs.g_b2, gs.g_l2 = ptr(blk[2].conv.weight.grad), ptr(blk[2].actnorm.bias.grad), ptr(blk[2].actnorm.logs.grad)
                gs.g_w3, gs.g_bias3, gs.g_l3 = ptr(blk[4].conv.weight.grad), ptr(blk[4].conv.bias.grad), ptr(blk[4].logs.grad)
                off, nb = part_off[r.idx]
                gs.mix_part = parts.data_ptr() + 4 * off
                gx = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                ws = ws_pair[n_step & 1]
                g_main, g_ext = (g, ext) if g is not None else (ext, None)
                nxt, nxt_p = None, None
                if i + 1 < len(order) and order[i + 1][0].kind == "step":
                    rn, svn = order[i + 1][0], order[i + 1][1]
                    if (rn.H, rn.W, rn.C) == (r.H, r.W, r.C):
                        nxt = capi.Step2dNext()
                        nxt.s = C.pointer(rn.cs)
                        nxt.z, nxt.hsave, nxt.g_ext = ptr(svn[1]), ptr(svn[4]), ptr(_ext(i + 1))
                        nxt.ws = ptr(ws_pair[(n_step + 1) & 1])
                        nxt.ga = (ga_all.data_ptr() + 4 * ga_off[rn.idx]) if deferred else None
                        nxt_p = C.byref(nxt)
                rows_out = C.c_int32(0)
                check(lib.nfk_step2d_bwd_chain(C.byref(r.cs), C.byref(gs), ptr(x_in), r.C, ptr(zb), ptr(h1), ptr(h2), ptr(hs),
                                               ptr(g_main), ptr(g_ext), ptr(g_ld), ptr(gx), ptr(ws),
                                               (ga_all.data_ptr() + 4 * ga_off[r.idx]) if deferred else None,
                                               ptr(wslab) if deferred else None, rows_in, nxt_p, C.byref(rows_out),
                                               C.byref(pend) if pend is not None else None, s),
                      "nfk_step2d_bwd_chain")
                rows_in = rows_out.value
                n_step += 1
                g = gx
                if eng.grad_hook is not None:
                    eng.grad_hook(r.idx)
            elif r.kind == "step":
                x_in, zb, h1, h2, hs, _, masks = sv
                blk = r.mod.block
                gs = Step2dGrads()
                gs.relu_masks = ptr(masks)
                gs.bwd_mode, gs.aux = eng.bwd_mode, eng.aux
                gs.g_w1, gs.g_b1, gs.g_l1 = ptr(blk[0].conv.weight.grad), ptr(blk[0].actnorm.bias.grad), ptr(blk[0].actnorm.logs.grad)
                gs.g_w2, gs.g_b2, gs.g_l2 = ptr(blk[2].conv.weight.grad), ptr(blk[2].actnorm.bias.grad), ptr(blk[2].actnorm.logs.grad)
                gs.g_w3, gs.g_bias3, gs.g_l3 = ptr(blk[4].conv.weight.grad), ptr(blk[4].conv.bias.grad), ptr(blk[4].logs.grad)
                off, nb = part_off[r.idx]
                gs.mix_part = parts.data_ptr() + 4 * off
                gx = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                (ws,) = scratch.get(lib.nfk_step2d_bwd_ws_floats(C.byref(r.cs)))
                g_main, g_ext = (g, ext) if g is not None else (ext, None)
                if deferred:
                    check(lib.nfk_step2d_bwd_deferred(C.byref(r.cs), C.byref(gs), ptr(x_in), r.C, ptr(zb), ptr(h1), ptr(h2),
                                                      ptr(hs), ptr(g_main), ptr(g_ext), ptr(g_ld), ptr(gx), ptr(ws),
                                                      ga_all.data_ptr() + 4 * ga_off[r.idx], ptr(wslab), s),
                          "nfk_step2d_bwd_deferred")
                else:
                    check(lib.nfk_step2d_bwd(C.byref(r.cs), C.byref(gs), ptr(x_in), r.C, ptr(zb), ptr(h1), ptr(h2), ptr(hs),
                                             ptr(g_main), ptr(g_ext), ptr(g_ld), ptr(gx), ptr(ws), s), "nfk_step2d_bwd")
                g = gx
                if eng.grad_hook is not None:
                    eng.grad_hook(r.idx)
            elif r.kind == "split":
                x_in, hs = sv
                cz = r.mod.conv
                gx = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                (ws,) = scratch.get(lib.nfk_split2d_bwd_ws_floats(C.byref(r.cs)))
                g_main, g_ext = (g, ext) if g is not None else (ext, None)
                check(lib.nfk_split2d_bwd(C.byref(r.cs), ptr(x_in), ptr(hs), ptr(g_main), ptr(g_ext), ptr(g_ld), ptr(gx),
                                          ptr(cz.conv.weight.grad), ptr(cz.conv.bias.grad), ptr(cz.logs.grad), ptr(ws), s),
                      "nfk_split2d_bwd")
                g = gx
                if eng.grad_hook is not None:
                    eng.grad_hook(r.idx)
            else:  # squeeze: grad of the input = unsqueeze of the grad
                first = r is recs[0]
                if first and not ctx.needs_input_grad[1]:
                    g = None
                    continue
                # (a squeeze output that is also a hint arrives as two gradients: summed inside the unsqueeze launch)
                g_a, g_b = (g, ext) if g is not None else (ext, None)
                if first and ctx.x_contig:
                    gx = torch.empty((B, r.C, r.H, r.W), device=dev, dtype=torch.float32)
                    check(lib.nfk_unsqueeze_add(ptr(g_a), ptr(g_b), ptr(gx), B, r.C, r.H, r.W, 1, s), "nfk_unsqueeze_add")
                    g = gx
                else:
                    gx = torch.empty((B, r.H, r.W, r.C), device=dev, dtype=torch.float32)
                    check(lib.nfk_unsqueeze_add(ptr(g_a), ptr(g_b), ptr(gx), B, r.C, r.H, r.W, 0, s), "nfk_unsqueeze_add")
                    g = gx if not first else gx.permute(0, 3, 1, 2)

        if pend is not None:
            check(lib.nfk_step2d_pending_flush(C.byref(pend), s), "nfk_step2d_pending_flush")
        for pd_, pg_, n_ in last_prep:
            check(lib.nfk_prep_bwd(pd_, pg_.data_ptr(), n_, ptr(g_ld), B, st["any_dense"], s), "nfk_prep_bwd (last level)")
        if deferred or aux_used:
            check(lib.nfk_aux_join(eng.aux, s), "nfk_aux_join")
        # parameter-side chain for all steps in one launch
        pg = []
        for r in recs:
            d = PrepGrad()
            if r.kind == "step" and r.idx not in prep_done:   # (prep_done: a level call has run this step's chain already)
                d = _prep_grad(r, nparts_used.get(r.idx, part_off[r.idx][1]))
            if r.kind in ("step", "split"):
                pg.append(d)
        sig = tuple((d.part, d.nparts, d.g_an_bias, d.g_lower, d.g_weight) for d in pg)
        if st["grad_sig"] != sig:
            st["grad_dev"], st["grad_sig"] = capi.struct_array_to_device(pg, dev), sig
        if any(r.kind == "step" and r.idx not in prep_done for r in recs):
            check(lib.nfk_prep_bwd(ptr(st["prep_dev"]), ptr(st["grad_dev"]), st["n_prep"], ptr(g_ld), B, st["any_dense"], s),
                  "nfk_prep_bwd")
        ctx.keep = (parts, keep_alive)
        gx = g if ctx.needs_input_grad[1] else None
        if gx is not None and gx.dim() == 4 and recs[0].kind != "squeeze":
            gx = gx.permute(0, 3, 1, 2)
        return (None, gx, g_ld if ctx.needs_input_grad[2] else None) + (None,) * (n_in - 3)


class _PriorFn(torch.autograd.Function):
    """objective = logdet + log N(z_top; prior) and bits/dim (models/kd_flows.py:136-150)."""

    @staticmethod
    def forward(ctx, z, prior, ld, scale):
        lib, s = capi.lib(), capi.stream()
        is_1d = z.dim() == 2
        zz = z.contiguous() if is_1d else _nhwc(z)
        B = zz.shape[0]
        Cc = zz.shape[-1]
        HW = 1 if is_1d else zz.shape[1] * zz.shape[2]
        pr = prior.to(torch.float32)
        bstride = 0 if (pr.shape[0] == 1 or pr.stride(0) == 0) else 2 * Cc * HW
        pr = (pr[:1] if bstride == 0 else pr).contiguous()
        obj = torch.empty(B, device=z.device, dtype=torch.float32)
        bpd = torch.empty(B, device=z.device, dtype=torch.float32)
        ld = ld.contiguous()
        check(lib.nfk_prior_fwd(ptr(zz), Cc, ptr(pr), bstride, ptr(ld), ptr(obj), ptr(bpd), B, HW, Cc, scale, s),
              "nfk_prior_fwd")
        ctx.meta = (zz, pr, bstride, B, HW, Cc, scale, is_1d, prior.shape)
        return bpd

    @staticmethod
    def backward(ctx, g_bpd):
        zz, pr, bstride, B, HW, Cc, scale, is_1d, pshape = ctx.meta
        lib, s = capi.lib(), capi.stream()
        gz = torch.empty_like(zz)
        gld = torch.empty(B, device=zz.device, dtype=torch.float32)
        gp = None
        if ctx.needs_input_grad[1]:
            gp = torch.empty((B, 2 * Cc) + tuple(pshape[2:]), device=zz.device, dtype=torch.float32)
        check(lib.nfk_prior_bwd(ptr(zz), Cc, ptr(pr), bstride, ptr(g_bpd.contiguous()), ptr(gz), Cc, ptr(gld), ptr(gp),
                                B, HW, Cc, scale, s), "nfk_prior_bwd")
        if gp is not None and pshape[0] == 1:
            gp = gp.sum(0, keepdim=True)
        return (gz if is_1d else gz.permute(0, 3, 1, 2)), gp, gld, None


def prior_bpd(z_last, prior_t, logdet, scale):
    return _PriorFn.apply(z_last, prior_t, logdet, float(scale))


def run_single_layer(layer, input, y_onehot, logdet, reverse, temperature=None, eps=None):
    """`layer(input, ...)` for a lone FlowStep / SqueezeLayer / Split2d, as the reference allows."""
    eng = getattr(layer, "_solo_engine", None)
    if eng is None:
        eng = FlowEngine([layer])
        object.__setattr__(layer, "_solo_engine", eng)
    if not reverse:
        ld_in = 0.0 if logdet is None else logdet
        outs, ld = eng.encode(input, y_onehot, ld_in)
        return outs[-1], (None if logdet is None else ld)
    # reverse: log-det bookkeeping of a lone layer is not needed by any caller of the hot path
    e = None if eps is None else {0: eps}
    out = eng.decode_single(input, y_onehot, temperature, e)
    return out, logdet
