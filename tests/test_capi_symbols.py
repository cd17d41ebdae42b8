"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every function
that include/nfk.h declares, and the ctypes prototypes cover all of them (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "nfk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nfk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_functions():
    names = declared_functions()
    assert "nfk_step2d_fwd" in names and "nfk_adam_step" in names and len(names) >= 25


def test_library_exports_every_declared_symbol():
    from nf_distillation_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(handle, n)]
    assert not missing, f"libnfk.so lacks {missing}"


def test_ctypes_prototypes_cover_header():
    from nf_distillation_amd import capi
    assert sorted(capi.exported_symbols()) == declared_functions()
    capi.lib()  # binds argtypes for every prototype; raises if a symbol is absent


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: a CPU tensor must raise, never silently compute."""
    import torch
    import nf_distillation_amd as nfa
    from nf_distillation_amd.capi import NfkUnavailable
    cfg = dict(image_shape=[8, 8, 3], hidden_channels=16, K=1, L=1, actnorm_scale=1.0,
               flow_permutation="invconv", flow_coupling="affine", LU_decomposed=True, y_classes=10,
               learn_top=False, y_condition=False, is_1d=False)
    m = nfa.create_glow_model(cfg)
    with pytest.raises(NfkUnavailable):
        m(torch.zeros(2, 3, 8, 8), None)
