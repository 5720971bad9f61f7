"""Drop-in for the reference's compiled operator module ``MultiScaleDeformableAttention``
(name from openvis/modeling/pixel_decoder/ops/setup.py:70; pybind surface vision.cpp:18-21;
imported at ops/functions/ms_deform_attn_func.py:21-29).

Same two functions, same argument order and the same error behaviour
(ms_deform_attn.h:26-67, cuda/ms_deform_attn_cuda.cu:33-57): contiguity / device checks
raise RuntimeError, CPU tensors raise "Not implemented on the CPU", fp32/fp64 only.
The computation is the hand-written gfx950 kernel behind ``ovis_msda_forward_f32/_f64``.
"""
import torch

from . import _lib


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")  # ms_deform_attn.h:43
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes),
                    ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                    ("attn_weight", attn_weight)):
        _require(t.is_contiguous(), f"{name} tensor has to be contiguous")   # cuda.cu:33-37
        _require(t.is_cuda, f"{name} must be a CUDA tensor")                 # cuda.cu:39-43
    _require(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
             "spatial_shapes / level_start_index must be int64")            # data<int64_t>(), cuda.cu:72-73
    _require(value.dtype in (torch.float32, torch.float64), "ms_deform_attn_forward_cuda not implemented for "
             f"'{value.dtype}'")                                             # AT_DISPATCH_FLOATING_TYPES, cuda.cu:69
    _require(sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype,
             "value / sampling_loc / attn_weight dtype mismatch")
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(batch, int(im2col_step))
    _require(batch % step == 0, f"batch({batch}) must divide im2col_step({step})")  # cuda.cu:57
    out = torch.empty((batch, num_query, num_heads * channels), dtype=value.dtype, device=value.device)
    fn = "ovis_msda_forward_f32" if value.dtype == torch.float32 else "ovis_msda_forward_f64"
    with torch.cuda.device(value.device):
        _lib.call(fn, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out,
                  batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
                  _lib.stream_ptr())
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    raise NotImplementedError(
        "ms_deform_attn_backward: openvis_amd covers the eval-only inference path (training kernels "
        "ms_deformable_col2im_* are out of scope, SURVEY.md §2b K2)")
