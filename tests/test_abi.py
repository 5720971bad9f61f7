"""CPU: the C-ABI library loads and exports every symbol include/openvis_hip.h declares;
argument validation happens before any device work (no compute calls without a GPU)."""
import ctypes
import os
import re

from tests.conftest import ROOT
from openvis_amd import _lib


def test_library_exports_every_declared_symbol():
    syms = _lib.declared_symbols()
    assert "ovis_msda_forward_f32" in syms and "ovis_last_error" in syms
    lib = _lib.lib()
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.ovis_abi_version() >= 1


def test_header_cites_reference_for_every_entry_point_group():
    src = open(os.path.join(ROOT, "include", "openvis_hip.h")).read()
    assert "ms_deform_im2col_cuda.cuh:242-304" in src and "vision.cpp:19" in src


def test_invalid_arguments_are_rejected_before_launch():
    lib = _lib.lib()
    n = ctypes.c_void_p(0)
    rc = lib.ovis_msda_forward_f32(n, n, n, n, n, n, 1, 1, 1, 1, 1, 1, 1, n)
    assert rc == 1  # OVIS_EINVAL
    assert b"null pointer" in lib.ovis_last_error()


def test_no_torch_types_in_abi_signatures():
    src = open(os.path.join(ROOT, "include", "openvis_hip.h")).read()
    assert not re.search(r"at::|torch::|Tensor", src)


def test_b1_is_a_compiled_torch_extension_with_dispatcher_ops():
    """B1: the module the reference imports (ops/functions/ms_deform_attn_func.py:22, built from
    src/vision.cpp:18-21) is a compiled extension with at::Tensor arguments, and the op is registered with the dispatcher."""
    import pytest
    import torch
    import MultiScaleDeformableAttention as MSDA
    assert MSDA.__file__.endswith(".so")
    assert callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    v, ss, lsi = torch.zeros(1, 6, 8, 32), torch.tensor([[2, 3]]), torch.tensor([0])
    loc, w = torch.zeros(1, 4, 8, 1, 4, 2), torch.zeros(1, 4, 8, 1, 4)
    for fn in (MSDA.ms_deform_attn_forward, torch.ops.ovis_mi.ms_deform_attn_forward):
        with pytest.raises(RuntimeError, match="Not implemented on the CPU"):          # ms_deform_attn.h:43
            fn(v, ss, lsi, loc, w, 64)
    with pytest.raises(NotImplementedError):
        MSDA.ms_deform_attn_backward(v, ss, lsi, loc, w, v, 64)
    m = lambda t: t.to("meta")                                                          # fake-tensor / torch.compile shape inference
    assert torch.ops.ovis_mi.ms_deform_attn_forward(m(v), m(ss), m(lsi), m(loc), m(w), 64).shape == (1, 4, 256)
