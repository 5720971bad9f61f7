"""GPU parity of K1 (ovis_msda_forward_*) through the drop-in operator module, against the
plain-C oracle (bit-exact) and the reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from oracle import msda as oracle_msda

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(GOLDEN, "msda.npz"))


def _run(value, shapes, lsi, loc, w, step=128):
    import MultiScaleDeformableAttention as MSDA  # the drop-in module, by the reference's import name
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = MSDA.ms_deform_attn_forward(t(value), t(shapes), t(lsi), t(loc), t(w), step)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_reference_fixture_double_and_float():
    o = _run(G["testpy_double_value"].astype(np.float64), G["testpy_shapes"], G["testpy_lsi"],
             G["testpy_double_loc"].astype(np.float64), G["testpy_double_w"].astype(np.float64), step=2)
    assert np.allclose(o, G["testpy_double_out"], rtol=1e-5, atol=1e-8)       # ops/test.py:43
    o = _run(G["testpy_float_value"], G["testpy_shapes"], G["testpy_lsi"], G["testpy_float_loc"],
             G["testpy_float_w"], step=2)
    assert np.allclose(o, G["testpy_float_out"], rtol=1e-2, atol=1e-3)        # ops/test.py:59


@pytest.mark.parametrize("name", ["enc", "oddD", "wide", "L4"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_bit_exact_vs_oracle_and_golden(name, dt):
    args = (G[f"{name}_shapes"], G[f"{name}_lsi"], G[f"{name}_loc"].astype(dt), G[f"{name}_w"].astype(dt))
    v = G[f"{name}_value"].astype(dt)
    o = _run(v, *args)
    ref = oracle_msda.msda_forward(v, *args)
    assert np.array_equal(o, ref), np.abs(o - ref).max()
    gold = G[f"{name}_out32" if dt == np.float32 else f"{name}_out64"]
    assert np.allclose(o, gold, rtol=1e-2, atol=1e-3)


def _encoder_inputs(B, sizes, seed=0, M=8, D=32, P=4):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(sizes, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    L = len(sizes)
    value = torch.randn(B, S, M, D, generator=g)
    # encoder-like: reference point of the token + bounded offsets (ms_deform_attn.py:106-109)
    ref = []
    for (H, W) in sizes:
        ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ref.append(torch.stack((xs.reshape(-1) / W, ys.reshape(-1) / H), -1))
    ref = torch.cat(ref, 0)
    off = torch.randn(B, S, M, L, P, 2, generator=g) * 2.0
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    w = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    return value.numpy(), shapes.numpy(), lsi.numpy(), loc.contiguous().numpy(), w.numpy()


def test_480p_encoder_shape_bit_exact():
    # BASELINE.json configs[0]: one 480p frame, S = 8505 tokens — the oracle finishes in seconds
    v, sh, lsi, loc, w = _encoder_inputs(1, [(15, 27), (30, 54), (60, 108)])
    o = _run(v, sh, lsi, loc, w)
    ref = oracle_msda.msda_forward(v, sh, lsi, loc, w)
    assert np.array_equal(o, ref)


def test_720p_full_size_properties():
    # BASELINE.json configs[1] size (5 frames, S = 19320): size-independent properties.
    v, sh, lsi, loc, w = _encoder_inputs(5, [(23, 40), (46, 80), (92, 160)], seed=1)
    o = _run(v, sh, lsi, loc, w)
    assert o.shape == (5, 19320, 256) and np.isfinite(o).all()
    # (1) per-frame independence + sampled bit-exactness against the oracle on frame 3
    ref3 = oracle_msda.msda_forward(v[3:4], sh, lsi, loc[3:4], w[3:4])
    assert np.array_equal(o[3:4], ref3)
    # (2) linearity in value: f(2v) == 2 f(v) exactly (power-of-two scaling commutes with rounding)
    o2 = _run(2.0 * v, sh, lsi, loc, w)
    assert np.array_equal(o2, 2.0 * o)
    # (3) constant value field => output == sum of weights of in-range taps <= 1
    oc = _run(np.ones_like(v), sh, lsi, loc, w)
    assert oc.max() <= 1.0 + 1e-5 and oc.min() >= 0.0
    # (4) batch chunking (im2col_step) has no numerical effect (cuda.cu:55-80)
    assert np.array_equal(_run(v, sh, lsi, loc, w, step=1), o)


def test_error_behaviour_matches_reference():
    import MultiScaleDeformableAttention as MSDA
    v, sh, lsi, loc, w = [torch.from_numpy(a) for a in _encoder_inputs(3, [(2, 2)], M=2, D=4)]
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):      # ms_deform_attn.h:43
        MSDA.ms_deform_attn_forward(v, sh, lsi, loc, w, 128)
    c = lambda t: t.cuda()
    with pytest.raises(RuntimeError, match="contiguous"):                       # cuda.cu:33
        MSDA.ms_deform_attn_forward(c(v).transpose(2, 3), c(sh), c(lsi), c(loc), c(w), 128)
    with pytest.raises(RuntimeError, match="must divide"):                      # cuda.cu:57
        MSDA.ms_deform_attn_forward(c(v), c(sh), c(lsi), c(loc), c(w), 2)
    with pytest.raises(RuntimeError):                                           # fp16 not dispatched, cuda.cu:69
        MSDA.ms_deform_attn_forward(c(v).half(), c(sh), c(lsi), c(loc).half(), c(w).half(), 128)
    with pytest.raises(NotImplementedError):
        MSDA.ms_deform_attn_backward(c(v), c(sh), c(lsi), c(loc), c(w), c(v), 128)


def test_reference_call_sequence_and_dispatcher_op_give_the_same_bits():
    """The reference-shaped call chain MSDeformAttnFunction.apply -> MSDA.ms_deform_attn_forward (func.py:34-39) and the
    dispatcher op torch.ops.ovis_mi.ms_deform_attn_forward run the same kernel on the caller's current stream."""
    import MultiScaleDeformableAttention as MSDA
    from openvis_amd.modeling.pixel_decoder.ops.functions.ms_deform_attn_func import MSDeformAttnFunction
    assert MSDA.__file__.endswith(".so")
    v, sh, lsi, loc, w = [torch.from_numpy(a).cuda() for a in _encoder_inputs(5, [(6, 9), (3, 5)], M=8, D=32)]
    a = MSDeformAttnFunction.apply(v, sh, lsi, loc, w, 128)
    b = torch.ops.ovis_mi.ms_deform_attn_forward(v, sh, lsi, loc, w, 128)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = MSDA.ms_deform_attn_forward(v, sh, lsi, loc, w, 128)
    side.synchronize()
    torch.cuda.synchronize()
    ref = oracle_msda.msda_forward(*[t.cpu().numpy() for t in (v, sh, lsi, loc, w)])
    for o in (a, b, c):
        assert np.array_equal(o.cpu().numpy(), ref)


@pytest.mark.parametrize("vdtype", ["bfloat16", "float16"])
def test_k1_16bit_value_variant_equals_the_f32_op_on_the_widened_values(vdtype):
    """SURVEY.md 8(b) "+ bf16-value variant": value stored as bf16 (or fp16), locations / weights / output f32.  The taps are widened exactly
    and the arithmetic is the f32 kernel's, so the result must equal oracle/msda_ref.c (and ovis_msda_forward_f32) on the widened value tensor
    BIT FOR BIT -- encoder shape at 480p, 2 frames, locations partly outside the maps."""
    import torch
    import MultiScaleDeformableAttention as MSDA
    g = torch.Generator().manual_seed(11)
    shapes = torch.tensor([(15, 27), (30, 54), (60, 107)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    dt = getattr(torch, vdtype)
    v16 = torch.randn(2, S, 8, 32, generator=g).to(dt)
    loc = torch.rand(2, S, 8, 3, 4, 2, generator=g) * 1.3 - 0.15
    w = torch.softmax(torch.randn(2, S, 8, 12, generator=g), -1).view(2, S, 8, 3, 4)
    out = MSDA.ms_deform_attn_forward(v16.cuda(), shapes.cuda(), lsi.cuda(), loc.cuda(), w.cuda(), 64)
    assert out.dtype == torch.float32
    wide = v16.float()
    ref = oracle_msda.msda_forward(wide.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), w.numpy())
    assert np.array_equal(out.cpu().numpy(), ref)
    out32 = MSDA.ms_deform_attn_forward(wide.cuda(), shapes.cuda(), lsi.cuda(), loc.cuda(), w.cuda(), 64)
    assert torch.equal(out, out32)
