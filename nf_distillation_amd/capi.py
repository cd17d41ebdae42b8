"""ctypes binding of libnfk.so (the C ABI declared in include/nfk.h).

There is no CPU fallback: importing the package works without the library (so that host-side
logic can be unit-tested), but the first use of any kernel raises ``NfkUnavailable`` if
``libnfk.so`` has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnfk.so")

c_fp = C.c_void_p  # device float*


class NfkUnavailable(RuntimeError):
    pass


class NfkError(RuntimeError):
    pass


class PrepDesc(C.Structure):
    _fields_ = [("C", C.c_int32), ("pixels", C.c_int32), ("is_1d", C.c_int32), ("perm_kind", C.c_int32),
                ("n1", C.c_int32), ("n2", C.c_int32), ("n3", C.c_int32), ("pad_", C.c_int32),
                ("an_bias", c_fp), ("an_logs", c_fp),
                ("lower", c_fp), ("upper", c_fp), ("log_s", c_fp), ("p", c_fp), ("sign_s", c_fp),
                ("weight", c_fp), ("perm_idx", c_fp),
                ("l1", c_fp), ("l2", c_fp), ("l3", c_fp),
                ("Wf", c_fp), ("cf", c_fp), ("Wi", c_fp), ("ci", c_fp), ("dld", c_fp),
                ("es1", c_fp), ("es2", c_fp), ("es3", c_fp), ("scratch", c_fp)]


class PrepGrad(C.Structure):
    _fields_ = [("part", c_fp), ("nparts", C.c_int32), ("pad_", C.c_int32),
                ("g_an_bias", c_fp), ("g_an_logs", c_fp), ("g_lower", c_fp), ("g_upper", c_fp),
                ("g_log_s", c_fp), ("g_weight", c_fp)]


class Step2d(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("hidden", C.c_int32), ("Cout", C.c_int32), ("additive", C.c_int32), ("bf16", C.c_int32),
                ("b1", c_fp), ("b2", c_fp), ("bias3", c_fp),
                ("Wf", c_fp), ("cf", c_fp), ("Wi", c_fp), ("ci", c_fp), ("dld", c_fp),
                ("es1", c_fp), ("es2", c_fp), ("es3", c_fp),
                ("wt1", c_fp), ("wt2", c_fp), ("wt3", c_fp), ("wd1", c_fp), ("wd2", c_fp), ("wd3", c_fp),
                ("wf1", c_fp), ("wf2", c_fp), ("wf3", c_fp), ("wg1", c_fp), ("wg2", c_fp), ("wg3", c_fp)]


class Step2dGrads(C.Structure):
    _fields_ = [("g_w1", c_fp), ("g_b1", c_fp), ("g_l1", c_fp), ("g_w2", c_fp), ("g_b2", c_fp),
                ("g_l2", c_fp), ("g_w3", c_fp), ("g_bias3", c_fp), ("g_l3", c_fp), ("mix_part", c_fp),
                ("relu_masks", c_fp), ("bwd_mode", C.c_int32), ("pad_", C.c_int32), ("aux", c_fp)]


class FusedRun(C.Structure):
    """nfk_fused_run (include/nfk.h): a run of FlowSteps of one level walked by ONE launch."""
    _fields_ = [("steps", c_fp), ("nsteps", C.c_int32), ("ldx", C.c_int32), ("x", c_fp), ("keep", C.c_uint64),
                ("zout", c_fp), ("zout_stride", C.c_int64), ("zsave", c_fp), ("zsave_stride", C.c_int64),
                ("h1save", c_fp), ("h2save", c_fp), ("h_stride", C.c_int64), ("hsave", c_fp), ("hsave_stride", C.c_int64),
                ("masksave", c_fp), ("mask_stride", C.c_int64), ("ldpart", c_fp)]


class LevelGrads(C.Structure):
    """nfk_level_grads: where the parameter gradients of one step of a level accumulate."""
    _fields_ = [("g_w1", c_fp), ("g_b1", c_fp), ("g_l1", c_fp), ("g_w2", c_fp), ("g_b2", c_fp),
                ("g_l2", c_fp), ("g_w3", c_fp), ("g_bias3", c_fp), ("g_l3", c_fp), ("mix_part", c_fp)]


class LevelBwd(C.Structure):
    """nfk_level_bwd (include/nfk.h): the backward of a run of FlowSteps of one level."""
    _fields_ = [("steps", c_fp), ("grads", c_fp), ("nsteps", C.c_int32), ("ldx0", C.c_int32), ("x0", c_fp),
                ("zout", c_fp), ("zout_stride", C.c_int64), ("zsave", c_fp), ("zsave_stride", C.c_int64),
                ("hsave", c_fp), ("hsave_stride", C.c_int64), ("h1save", c_fp), ("h2save", c_fp), ("h_stride", C.c_int64),
                ("masks", c_fp), ("mask_stride", C.c_int64), ("g_out", c_fp), ("g_ext", c_fp), ("g_ld", c_fp),
                ("g_x", c_fp), ("ws", c_fp), ("k_lo", C.c_int32), ("k_hi", C.c_int32), ("multi", C.c_int32),
                ("pad_", C.c_int32), ("aux", c_fp), ("wgrad_chunk", C.c_int32), ("pad2_", C.c_int32),
                ("prep_descs", c_fp), ("prep_grads", c_fp), ("prep_n", C.c_int32), ("prep_dense", C.c_int32)]


class Step2dPending(C.Structure):
    """nfk_step2d_pending: the slab / partial-row reductions a merged weight-gradient launch leaves to the next call."""
    _fields_ = [("valid", C.c_int32), ("pad_", C.c_int32), ("blob", C.c_uint64 * 40)]


class Step2dNext(C.Structure):
    _fields_ = [("s", C.POINTER(Step2d)), ("z", c_fp), ("hsave", c_fp), ("g_ext", c_fp), ("ws", c_fp), ("ga", c_fp)]


LD_MAX_PARTS = 8


class LdParts(C.Structure):
    _fields_ = [("part", c_fp), ("nsteps", C.c_int32), ("tiles", C.c_int32)]


class Split2d(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("bias", c_fp), ("es", c_fp), ("wt", c_fp), ("wd", c_fp), ("w", c_fp)]


class TabStepC(C.Structure):
    _fields_ = [("B", C.c_int32), ("d", C.c_int32), ("hid", C.c_int32), ("ncond", C.c_int32),
                ("additive", C.c_int32), ("made", C.c_int32),
                ("W", c_fp * 6), ("b", c_fp * 6),
                ("Wf", c_fp), ("cf", c_fp), ("Wi", c_fp), ("ci", c_fp), ("dld", c_fp)]


class TabGrads(C.Structure):
    _fields_ = [("gW", c_fp * 6), ("gb", c_fp * 6)]


KD_MAX_PAIRS = 16   # NFK_KD_MAX_PAIRS
# nfk_step2d_grads.bwd_mode (include/nfk.h)
BWD_DEFAULT, BWD_SEPARATE, BWD_DUAL_SMALL, BWD_DUAL, BWD_FUSED = 0, 1, 2, 3, 4
# nfk_tune knobs
(TUNE_F32_TILES, TUNE_BF16_BIG_TILE_MIN_K64, TUNE_WGRAD_MAX_SPLITS, TUNE_WGRAD_TARGET_WGS, TUNE_WGRAD_WIDE_MAX_SPLITS,
 TUNE_NARROW_HALO_DGRAD, TUNE_WGRAD_HALO, TUNE_WGRAD_HALO_MAX_SPLITS, TUNE_RUN_HALF_TILES, TUNE_SPLIT_FUSED, TUNE_TAB_RESIDENT) = range(11)


class KdPairs(C.Structure):
    _fields_ = [("n", C.c_int32), ("B", C.c_int32), ("divisor", C.c_int32), ("pad_", C.c_int32), ("weights", c_fp),
                ("s", c_fp * KD_MAX_PAIRS), ("t", c_fp * KD_MAX_PAIRS), ("per_sample", C.c_int64 * KD_MAX_PAIRS)]


class KdGrads(C.Structure):
    _fields_ = [("g_s", c_fp * KD_MAX_PAIRS), ("g_nll", c_fp)]


class RepackDesc(C.Structure):
    _fields_ = [("w", c_fp), ("wt", c_fp), ("wd", c_fp), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("taps", C.c_int32), ("frag", C.c_int32), ("wf", c_fp), ("wg", c_fp), ("gfrag", C.c_int32),
                ("flags", C.c_int32)]


REPACK_FRAG_ONLY = 1   # NFK_REPACK_FRAG_ONLY


i32, f32, u64, vp, lng = C.c_int, C.c_float, C.c_size_t, C.c_void_p, C.c_long

_PROTOS = {
    "nfk_version": (C.c_char_p, []),
    "nfk_profile_begin": (None, []),
    "nfk_tune": (None, [i32, i32]),
    "nfk_aux_create": (vp, [vp]),
    "nfk_aux_destroy": (None, [vp]),
    "nfk_profile_end": (None, [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]),
    "nfk_padded_k": (i32, [i32]),
    "nfk_conv_splitk": (i32, [i32, i32, i32]),
    "nfk_chanmix_blocks": (i32, [i32, i32]),
    "nfk_step2d_fwd_ws_floats": (u64, [C.POINTER(Step2d)]),
    "nfk_step2d_bwd_ws_floats": (u64, [C.POINTER(Step2d)]),
    "nfk_split2d_fwd_ws_floats": (u64, [C.POINTER(Split2d)]),
    "nfk_split2d_bwd_ws_floats": (u64, [C.POINTER(Split2d)]),
    "nfk_split2d_fused_ok": (i32, [C.POINTER(Split2d)]),
    "nfk_split2d_fwd_tiles": (i32, [C.POINTER(Split2d)]),
    "nfk_ld_finish_multi": (i32, [vp, C.POINTER(LdParts), i32, i32, vp, vp]),
    "nfk_prep": (i32, [vp, i32, i32, vp]),
    "nfk_prep_bwd": (i32, [vp, vp, i32, vp, i32, i32, vp]),
    "nfk_prep_inv_bwd": (i32, [vp, vp, i32, vp, i32, vp]),
    "nfk_repack": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "nfk_repack_batch": (i32, [vp, i32, vp]),
    "nfk_repack_batch_bf16": (i32, [vp, i32, vp]),
    "nfk_padded_k_bf16": (i32, [i32]),
    "nfk_prep_repack": (i32, [vp, i32, i32, vp, i32, vp, i32, vp]),
    "nfk_chanmix": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, vp]),
    "nfk_conv": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp, vp]),
    "nfk_coupling_fwd": (i32, [vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nfk_coupling_inv": (i32, [vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nfk_squeeze": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nfk_unsqueeze": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "nfk_unsqueeze_add": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nfk_prior_fwd": (i32, [vp, i32, vp, lng, vp, vp, vp, i32, i32, i32, f32, vp]),
    "nfk_prior_bwd": (i32, [vp, i32, vp, lng, vp, vp, i32, vp, vp, i32, i32, i32, f32, vp]),
    "nfk_step2d_fwd": (i32, [C.POINTER(Step2d), vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nfk_step2d_inv": (i32, [C.POINTER(Step2d), vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nfk_step2d_inv_bwd": (i32, [C.POINTER(Step2d), C.POINTER(Step2dGrads), vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nfk_step2d_bwd": (i32, [C.POINTER(Step2d), C.POINTER(Step2dGrads), vp, i32, vp, vp, vp, vp, vp, vp, vp, vp,
                             vp, vp]),
    "nfk_step2d_bwd_deferred": (i32, [C.POINTER(Step2d), C.POINTER(Step2dGrads), vp, i32, vp, vp, vp, vp, vp, vp, vp,
                                      vp, vp, vp, vp, vp]),
    "nfk_step2d_bwd_chain": (i32, [C.POINTER(Step2d), C.POINTER(Step2dGrads), vp, i32, vp, vp, vp, vp, vp, vp, vp,
                                   vp, vp, vp, vp, i32, C.POINTER(Step2dNext), C.POINTER(i32), C.POINTER(Step2dPending), vp]),
    "nfk_step2d_pending_flush": (i32, [C.POINTER(Step2dPending), vp]),
    "nfk_step2d_bwd_ga_floats": (u64, [C.POINTER(Step2d)]),
    "nfk_step2d_bwd_wslab_floats": (u64, [C.POINTER(Step2d)]),
    "nfk_aux_join": (i32, [vp, vp]),
    "nfk_level_bwd_ok": (i32, [C.POINTER(Step2d)]),
    "nfk_level_bwd_parts": (i32, [C.POINTER(Step2d)]),
    "nfk_level_bwd_ws_floats": (u64, [C.POINTER(Step2d), i32]),
    "nfk_level2d_bwd": (i32, [C.POINTER(LevelBwd), C.POINTER(Step2d), vp]),
    "nfk_step2d_fused_bwd_ok": (i32, [C.POINTER(Step2d)]),
    "nfk_fused_bwd_set_stamps": (None, [vp]),
    "nfk_debug_triple_parts": (None, [i32]),
    "nfk_step2d_mask_bytes": (u64, [C.POINTER(Step2d)]),
    "nfk_step2d_fused_tiles": (i32, [C.POINTER(Step2d)]),
    "nfk_fused_set_stamps": (None, [vp]),
    "nfk_step2d_fwd_fused": (i32, [C.POINTER(Step2d), vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "nfk_step2d_fused_ws_floats": (u64, [C.POINTER(Step2d)]),
    "nfk_run2d_fused_ok": (i32, [C.POINTER(Step2d)]),
    "nfk_run2d_fwd_fused": (i32, [C.POINTER(FusedRun), C.POINTER(Step2d), vp]),
    "nfk_ld_finish": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "nfk_split2d_fwd": (i32, [C.POINTER(Split2d), vp, vp, vp, vp, vp, vp, vp]),
    "nfk_split2d_inv": (i32, [C.POINTER(Split2d), vp, vp, f32, vp, vp, vp]),
    "nfk_split2d_bwd": (i32, [C.POINTER(Split2d), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nfk_tab_bwd_blocks": (i32, [C.POINTER(TabStepC)]),
    "nfk_tab_param_floats": (i32, [C.POINTER(TabStepC)]),
    "nfk_tab_fwd": (i32, [C.POINTER(TabStepC), vp, vp, vp, vp, vp, vp]),
    "nfk_tab_inv": (i32, [C.POINTER(TabStepC), vp, vp, vp, vp, vp, vp]),
    "nfk_tab_bwd": (i32, [C.POINTER(TabStepC), C.POINTER(TabGrads), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nfk_tab_inv_bwd": (i32, [C.POINTER(TabStepC), C.POINTER(TabGrads), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nfk_sumsq": (i32, [vp, u64, vp, vp, vp]),
    "nfk_preprocess_u8": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "nfk_postprocess_u8": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nfk_kd_loss_fwd": (i32, [C.POINTER(KdPairs), vp, f32, f32, vp, vp, vp, vp]),
    "nfk_kd_loss_bwd": (i32, [C.POINTER(KdPairs), vp, f32, f32, C.POINTER(KdGrads), vp]),
    "nfk_adam_step": (i32, [vp, vp, vp, vp, u64, vp, f32, f32, f32, f32, f32, f32, f32, i32, vp]),
    "nfk_adam_step_dev": (i32, [vp, vp, vp, vp, u64, vp, f32, f32, f32, f32, f32, f32, f32, vp, vp]),
}

_lib = None


def lib():
    """The loaded library; raises NfkUnavailable if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NfkUnavailable(
                f"{LIB_PATH} not found: build the HIP library first "
                "(`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


PROF_CLASSES = ("conv_gemm_kernel<4,4,2,2> (128x128)", "conv_gemm_kernel<2,2,2,2> (64x64)",
                "conv_gemm_kernel<*,2,1,4> (narrow-N)", "wgrad_kernel<4,4,2,2>", "wgrad_kernel<*,2,1,4>",
                "fused2_step_fwd_kernel<*,1,*> (FlowStep forward, Cout <= 16: CIFAR level 1 = <3,1,4,1,false,4>)",
                "fused2_step_fwd_kernel<*,2,*> (FlowStep forward, Cout <= 32: CIFAR level 2 = <2,2,8,1,true,4>: one launch per "
                "run of steps, flops and bytes of the whole run)",
                "fused2_step_fwd_kernel<*,3,*> (FlowStep forward, Cout <= 48: CIFAR level 3 = <1,3,16,1,true,1>, 16-pixel tiles: one launch per "
                "run of steps, flops and bytes of the whole run)",
                ) + tuple(f"dual_dgrad_wgrad_bf16_kernel {conv}, {lvl}"
                          for conv in ("conv3 (block.4: wide dgrad + narrow wgrad)", "conv2 (block.2: 512x512)",
                                       "conv1 (block.0: narrow dgrad + wide wgrad)")
                          for lvl in ("M>=8192 (CIFAR level 1)", "M>=2048 (level 2)", "M<2048 (level 3)")) \
    + tuple(f"{k}, {lvl}" for k in ("fusedb_step_bwd_kernel (FlowStep backward: dgrad3 -> dgrad2 -> dgrad1 [-> channel mix -> coupling "
                                    "backward of the step below, level mode] in one launch; a run of steps in one launch: flops and "
                                    "bytes of the whole run)",
                                    "level_wgrad_kernel / wgrad_tail_kernel (weight-gradient GEMMs: all steps of a level in one launch "
                                    "[level mode], or one step's three + channel mix + the previous step's reductions)")
            for lvl in ("M>=8192 (CIFAR level 1)", "M>=2048 (level 2)", "M<2048 (level 3)"))


# class -> rocprofv3 symbol of the instance the student's chain launches at CIFAR B=64 (tools/pmc_traffic.py, bench.py `traffic`)
PROF_SYMBOLS = {
    PROF_CLASSES[5]: "nfk::fused2_step_fwd_kernel<3, 1, 4, 1, false, 4>",
    PROF_CLASSES[6]: "nfk::fused2_step_fwd_kernel<2, 2, 8, 1, true, 4>",
    PROF_CLASSES[7]: "nfk::fused2_step_fwd_kernel<1, 3, 16, 1, true, 1>",
    PROF_CLASSES[17]: "nfk::fusedb_step_bwd_kernel<3, 1, 8, 4, true, false>",
    PROF_CLASSES[18]: "nfk::fusedb_step_bwd_kernel<2, 1, 16, 4, true, true>",
    PROF_CLASSES[19]: "nfk::fusedb_step_bwd_kernel<1, 2, 28, 1, true, true>",
    PROF_CLASSES[20]: "nfk::level_wgrad_kernel",
}


def profile_begin():
    lib().nfk_profile_begin()


def profile_end():
    """-> list of dicts per kernel class: ms, launches, flops, bytes."""
    n = len(PROF_CLASSES)
    ms, la, fl, by = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)(), (C.c_double * n)()
    lib().nfk_profile_end(ms, la, fl, by)
    return [dict(kernel=PROF_CLASSES[i], ms=ms[i], launches=la[i], flops=fl[i], bytes=by[i]) for i in range(n)]


def exported_symbols():
    return sorted(_PROTOS)


def check(rc, what="nfk call"):
    if rc != 0:
        raise NfkError(f"{what} failed with status {rc}")


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def require_cuda(t, what="input"):
    if not t.is_cuda:
        raise NfkUnavailable(f"{what} is on {t.device}: the nfk kernels run on MI355X only (no CPU fallback)")


def struct_array_to_device(items, device):
    """Upload a list of ctypes structs as a contiguous device byte tensor."""
    if not items:
        return torch.empty(0, dtype=torch.uint8, device=device)
    arr = (type(items[0]) * len(items))(*items)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)
