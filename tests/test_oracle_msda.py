"""CPU: pin the plain-C K1 oracle (oracle/msda_ref.c) against golden vectors produced by the
reference's own ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:52-72)."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from oracle import msda

G = np.load(os.path.join(GOLDEN, "msda.npz"))


def test_reference_fixture_double():
    # ops/test.py:35-47 — fp64, torch.allclose default tolerances (rtol 1e-5, atol 1e-8)
    o = msda.msda_forward(G["testpy_double_value"].astype(np.float64), G["testpy_shapes"], G["testpy_lsi"],
                          G["testpy_double_loc"].astype(np.float64), G["testpy_double_w"].astype(np.float64))
    assert np.allclose(o, G["testpy_double_out"], rtol=1e-5, atol=1e-8)
    assert np.abs(o - G["testpy_double_out"]).max() < 1e-15


def test_reference_fixture_float():
    # ops/test.py:51-63 — fp32, rtol 1e-2 / atol 1e-3
    o = msda.msda_forward(G["testpy_float_value"], G["testpy_shapes"], G["testpy_lsi"], G["testpy_float_loc"],
                          G["testpy_float_w"])
    assert np.allclose(o, G["testpy_float_out"], rtol=1e-2, atol=1e-3)
    assert np.abs(o - G["testpy_float_out"]).max() < 1e-8


@pytest.mark.parametrize("name", ["enc", "oddD", "wide", "L4"])
def test_model_shaped_cases(name):
    args = (G[f"{name}_shapes"], G[f"{name}_lsi"], G[f"{name}_loc"], G[f"{name}_w"])
    o32 = msda.msda_forward(G[f"{name}_value"], *args)
    o64 = msda.msda_forward(G[f"{name}_value"].astype(np.float64), *args)
    assert np.abs(o64 - G[f"{name}_out64"]).max() < 1e-13          # same math, fp64
    assert np.allclose(o32, G[f"{name}_out32"], rtol=1e-2, atol=1e-3)  # the reference's fp32 tolerance
    assert np.abs(o32 - G[f"{name}_out32"]).max() < 5e-6           # summation-order rounding only


def test_out_of_range_points_contribute_zero():
    shapes = np.array([[3, 4]], np.int64)
    lsi = np.array([0], np.int64)
    value = np.ones((1, 12, 1, 4), np.float32)
    loc = np.array([-1.0, -1.0, 2.0, 2.0, 0.5, 0.5], np.float32).reshape(1, 1, 1, 1, 3, 2)
    w = np.array([0.25, 0.25, 0.5], np.float32).reshape(1, 1, 1, 1, 3)
    o = msda.msda_forward(value, shapes, lsi, loc, w)
    assert np.array_equal(o, np.full((1, 1, 4), 0.5, np.float32))
