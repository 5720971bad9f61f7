// Compiled torch extension `MultiScaleDeformableAttention` (B1 of SURVEY.md 8(b)): the module the reference imports
// (ops/functions/ms_deform_attn_func.py:21-29; built by ops/setup.py:70 from src/vision.cpp:18-21), re-built for MI355X on top
// of the C ABI of libopenvis_hip.so.  Same two pybind functions with at::Tensor arguments, same checks and error texts
// (ms_deform_attn.h:26-67, cuda/ms_deform_attn_cuda.cu:33-57, 69), plus dispatcher ops `ovis_mi::ms_deform_attn_forward`
// / `_backward` (TORCH_LIBRARY) with a Meta kernel, so the op is visible to torch.compile / fake tensors.
//
// No arithmetic here: the kernel is csrc/msda.hip behind ovis_msda_forward_f32 / _f64, launched on the CURRENT stream of
// the calling thread like the reference (cuda.cu:70).  backward: the eval-only tier does not build the col2im kernels.
#include <torch/extension.h>
#include <torch/library.h>
// ROCm builds of torch present the HIP device under the "cuda" device type: guards and streams are the *MasqueradingAsCUDA forms
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include "../../../include/openvis_hip.h"

namespace {

void check_forward_args(const at::Tensor& value, const at::Tensor& spatial_shapes, const at::Tensor& level_start_index,
                        const at::Tensor& sampling_loc, const at::Tensor& attn_weight, int64_t im2col_step) {
  TORCH_CHECK(value.is_contiguous(), "value tensor has to be contiguous");                            // cuda.cu:33-37
  TORCH_CHECK(spatial_shapes.is_contiguous(), "spatial_shapes tensor has to be contiguous");
  TORCH_CHECK(level_start_index.is_contiguous(), "level_start_index tensor has to be contiguous");
  TORCH_CHECK(sampling_loc.is_contiguous(), "sampling_loc tensor has to be contiguous");
  TORCH_CHECK(attn_weight.is_contiguous(), "attn_weight tensor has to be contiguous");
  TORCH_CHECK(value.is_cuda(), "value must be a CUDA tensor");                                        // cuda.cu:39-43
  TORCH_CHECK(spatial_shapes.is_cuda(), "spatial_shapes must be a CUDA tensor");
  TORCH_CHECK(level_start_index.is_cuda(), "level_start_index must be a CUDA tensor");
  TORCH_CHECK(sampling_loc.is_cuda(), "sampling_loc must be a CUDA tensor");
  TORCH_CHECK(attn_weight.is_cuda(), "attn_weight must be a CUDA tensor");
  TORCH_CHECK(spatial_shapes.scalar_type() == at::kLong && level_start_index.scalar_type() == at::kLong,
              "spatial_shapes / level_start_index must be int64");                                    // data<int64_t>(), cuda.cu:72-73
  // MI355X extension of the dtype set (AT_DISPATCH_FLOATING_TYPES, cuda.cu:69): a bf16 / fp16 VALUE tensor with f32 locations and weights
  const bool v16 = value.scalar_type() == at::kBFloat16 || value.scalar_type() == at::kHalf;
  TORCH_CHECK(value.scalar_type() == at::kFloat || value.scalar_type() == at::kDouble || v16,
              "\"ms_deform_attn_forward_cuda\" not implemented for '", toString(value.scalar_type()), "'");   // cuda.cu:69
  const auto ctl = v16 ? at::kFloat : value.scalar_type();
  TORCH_CHECK(sampling_loc.scalar_type() == ctl && attn_weight.scalar_type() == ctl,
              "value / sampling_loc / attn_weight dtype mismatch (a bf16 / fp16 value goes with float32 locations and weights)");
  TORCH_CHECK(value.dim() == 4 && spatial_shapes.dim() == 2 && sampling_loc.dim() == 6 && attn_weight.dim() == 5,
              "ms_deform_attn_forward: value [N,S,M,D], spatial_shapes [L,2], sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P]");
  const int64_t batch = value.size(0);
  const int64_t step = std::min<int64_t>(batch, im2col_step);
  TORCH_CHECK(step > 0 && batch % step == 0, "batch(", batch, ") must divide im2col_step(", step, ")");   // cuda.cu:57
}

at::Tensor ms_deform_attn_forward(const at::Tensor& value, const at::Tensor& spatial_shapes, const at::Tensor& level_start_index,
                                  const at::Tensor& sampling_loc, const at::Tensor& attn_weight, int64_t im2col_step) {
  TORCH_CHECK(value.is_cuda(), "Not implemented on the CPU");                                          // ms_deform_attn.h:43
  check_forward_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step);
  const int batch = (int)value.size(0), spatial_size = (int)value.size(1), num_heads = (int)value.size(2), channels = (int)value.size(3);
  const int num_levels = (int)spatial_shapes.size(0), num_query = (int)sampling_loc.size(1), num_point = (int)sampling_loc.size(4);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
  at::Tensor out = at::empty({batch, num_query, (int64_t)num_heads * channels}, sampling_loc.options());      // fully overwritten (cuda.cu:59 zero-inits)
  hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  int rc;
  if (value.scalar_type() == at::kBFloat16 || value.scalar_type() == at::kHalf) {
    auto fn = value.scalar_type() == at::kBFloat16 ? ovis_msda_forward_bf16v : ovis_msda_forward_f16v;
    rc = fn(value.data_ptr(), spatial_shapes.data_ptr<int64_t>(), level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr<float>(),
            attn_weight.data_ptr<float>(), out.data_ptr<float>(), batch, spatial_size, num_heads, channels, num_levels, num_query, num_point,
            (ovis_stream_t)stream);
  } else if (value.scalar_type() == at::kFloat)
    rc = ovis_msda_forward_f32(value.data_ptr<float>(), spatial_shapes.data_ptr<int64_t>(), level_start_index.data_ptr<int64_t>(),
                               sampling_loc.data_ptr<float>(), attn_weight.data_ptr<float>(), out.data_ptr<float>(), batch, spatial_size,
                               num_heads, channels, num_levels, num_query, num_point, (ovis_stream_t)stream);
  else
    rc = ovis_msda_forward_f64(value.data_ptr<double>(), spatial_shapes.data_ptr<int64_t>(), level_start_index.data_ptr<int64_t>(),
                               sampling_loc.data_ptr<double>(), attn_weight.data_ptr<double>(), out.data_ptr<double>(), batch,
                               spatial_size, num_heads, channels, num_levels, num_query, num_point, (ovis_stream_t)stream);
  TORCH_CHECK(rc == OVIS_OK, "ms_deform_attn_forward: ", ovis_last_error());
  return out;
}

at::Tensor ms_deform_attn_forward_meta(const at::Tensor& value, const at::Tensor& spatial_shapes, const at::Tensor& level_start_index,
                                       const at::Tensor& sampling_loc, const at::Tensor& attn_weight, int64_t im2col_step) {
  return at::empty_symint({value.sym_size(0), sampling_loc.sym_size(1), value.sym_size(2) * value.sym_size(3)}, sampling_loc.options());
}

std::vector<at::Tensor> ms_deform_attn_backward(const at::Tensor& value, const at::Tensor& spatial_shapes, const at::Tensor& level_start_index,
                                                const at::Tensor& sampling_loc, const at::Tensor& attn_weight, const at::Tensor& grad_output,
                                                int64_t im2col_step) {
  TORCH_CHECK_NOT_IMPLEMENTED(false, "ms_deform_attn_backward: this build covers the eval-only inference path (the training kernels "
                                     "ms_deformable_col2im_* are out of scope, SURVEY.md 2b K2)");
}

}  // namespace

TORCH_LIBRARY(ovis_mi, m) {
  m.def("ms_deform_attn_forward(Tensor value, Tensor spatial_shapes, Tensor level_start_index, Tensor sampling_loc, "
        "Tensor attn_weight, int im2col_step) -> Tensor");
  m.def("ms_deform_attn_backward(Tensor value, Tensor spatial_shapes, Tensor level_start_index, Tensor sampling_loc, "
        "Tensor attn_weight, Tensor grad_output, int im2col_step) -> Tensor[]");
}
TORCH_LIBRARY_IMPL(ovis_mi, CUDA, m) {     // "CUDA" is the HIP device's dispatch key on ROCm builds of torch
  m.impl("ms_deform_attn_forward", &ms_deform_attn_forward);
  m.impl("ms_deform_attn_backward", &ms_deform_attn_backward);
}
TORCH_LIBRARY_IMPL(ovis_mi, CPU, m) {      // same texts as the reference's CPU stub (cpu/ms_deform_attn_cpu.cpp:31, ms_deform_attn.h:43)
  m.impl("ms_deform_attn_forward", &ms_deform_attn_forward);
  m.impl("ms_deform_attn_backward", &ms_deform_attn_backward);
}
TORCH_LIBRARY_IMPL(ovis_mi, Meta, m) {
  m.impl("ms_deform_attn_forward", &ms_deform_attn_forward_meta);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {   // vision.cpp:18-21
  m.def("ms_deform_attn_forward", &ms_deform_attn_forward, "ms_deform_attn_forward");
  m.def("ms_deform_attn_backward", &ms_deform_attn_backward, "ms_deform_attn_backward");
}
