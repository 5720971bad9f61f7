// Dispatcher ops `ovis_mi::*` for the hot stages of the path (north star: "called from Python through PyTorch-ROCm custom ops").
// The reference registers its one native operator through a torch extension (ops/src/vision.cpp:18-21); this file gives the
// stages that have no native counterpart there -- they are cuBLAS / ATen calls and Python loops in the reference -- the same
// standing: each op has a schema, a CUDA(HIP) kernel that forwards to the C ABI of libopenvis_hip.so on the CURRENT stream of the
// calling thread, and a Meta kernel (output shapes / dtypes only), so `torch.compile`, fake tensors and the profiler see them.
// openvis_amd/ops.py calls THESE (torch.ops.ovis_mi.<name>) -- one path; the ctypes binding stays for the long tail of small kernels.
//
//   gemm_nt_f16        the CLIP ViT GEMMs (mask_adapted_clip/model.py:238-268; fp16 on the reference's GPU path, adapter.py:108-111)
//   gemm_nt_f16_ln     ln_1 -> in_proj / ln_2 -> c_fc with the LayerNorm folded into the GEMM (model.py:262-267)
//   gemm_nt_f16_res16_stats  out_proj / c_proj on the fp16 stream + partial LayerNorm statistics of the rows it writes (row_stats_finalize)
//   row_stats_f16      (mean, rstd) of fp16 rows (the statistics half of model.py:157-163)
//   msda_encoder_fused softmax + sampling locations + K1 (ms_deform_attn.py:102-118 + ms_deform_im2col_cuda.cuh:242-304)
//   attention_f16      nn.MultiheadAttention core of the CLIP blocks (model.py:254-263)
//   mask_bbox          boxes of {sigmoid(x4 upsample) > .5} (openvis.py:87-96, adapter.py:88-94)
//   clip_crop_patches  roi_align crops written as the patch-embedding matrix (adapter.py:96-116, 140-143)
//   hungarian_link     the MinVIS tracker chain (minvis.py:28-72)
//   topk_entropy       top-10 over [Q_valid x K] + entropies (video_maskformer.py:267-278)
//
// No arithmetic here.
#include <torch/extension.h>
#include <torch/library.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include "../../../include/openvis_hip.h"

namespace {

ovis_stream_t cur_stream() { return (ovis_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

void need(const at::Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "ovis_mi: ", what, " must be a contiguous HIP tensor (no CPU fallback)");
}
const float* fptr(const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }

// ---- gemm_nt_f16 ---------------------------------------------------------------------------------------------------
at::Tensor gemm_nt_f16(const at::Tensor& a, const at::Tensor& w, const c10::optional<at::Tensor>& bias,
                       const c10::optional<at::Tensor>& residual, int64_t act, bool out_f16) {
  need(a, "a"); need(w, "w");
  TORCH_CHECK(a.dim() == 2 && w.dim() == 2 && a.size(1) == w.size(1), "gemm_nt_f16: a [M,K], w [N,K]");
  TORCH_CHECK(a.scalar_type() == at::kHalf && w.scalar_type() == at::kHalf, "gemm_nt_f16 needs fp16 operands");
  const int M = (int)a.size(0), K = (int)a.size(1), N = (int)w.size(0);
  if (bias.has_value() && bias->defined()) { need(*bias, "bias"); TORCH_CHECK(bias->scalar_type() == at::kFloat && bias->numel() == N, "gemm_nt_f16: bias f32 [N]"); }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
  const bool has_r = residual.has_value() && residual->defined();
  if (has_r) { need(*residual, "residual"); TORCH_CHECK(residual->dim() == 2 && residual->size(0) == M && residual->size(1) == N, "gemm_nt_f16: residual [M,N]"); }
  int rc;
  if (has_r && residual->scalar_type() == at::kHalf) {
    // the tower's fp16 residual stream: C (fp16) = A B^T + bias + R (fp16), accumulated in f32 (ping-pong kernel shapes only)
    TORCH_CHECK(act == 0, "gemm_nt_f16: an fp16 residual goes with act = none (out-proj / c_proj)");
    at::Tensor out = at::empty({M, N}, a.options());
    TORCH_CHECK(ovis_gemm_nt_f16_res16_eligible(out.data_ptr(), residual->data_ptr(), K, K, N, N, M, N, K, fptr(bias)) != 0,
                "gemm_nt_f16: fp16 residual on a shape the ping-pong kernel does not take (M=", M, " N=", N, " K=", K,
                "): pass an f32 residual");
    rc = ovis_gemm_nt_f16_res16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, fptr(bias), residual->data_ptr(), N, cur_stream());
    TORCH_CHECK(rc == OVIS_OK, "gemm_nt_f16: ", ovis_last_error());
    return out;
  }
  TORCH_CHECK(!has_r || residual->scalar_type() == at::kFloat, "gemm_nt_f16: residual must be f32 or fp16");
  at::Tensor out = at::empty({M, N}, a.options().dtype(out_f16 ? at::kHalf : at::kFloat));
  rc = ovis_gemm_nt_f16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, fptr(bias), fptr(residual), N, (int)act, out_f16 ? 1 : 0,
                        cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "gemm_nt_f16: ", ovis_last_error());
  return out;
}
at::Tensor gemm_nt_f16_meta(const at::Tensor& a, const at::Tensor& w, const c10::optional<at::Tensor>&,
                            const c10::optional<at::Tensor>& residual, int64_t, bool out_f16) {
  const bool r16 = residual.has_value() && residual->defined() && residual->scalar_type() == at::kHalf;
  return at::empty_symint({a.sym_size(0), w.sym_size(0)}, a.options().dtype(out_f16 || r16 ? at::kHalf : at::kFloat));
}

// ---- LayerNorm folded into the GEMM (include/openvis_hip.h: ovis_gemm_nt_f16_ln, ovis_row_stats_f16, ovis_gemm_nt_f16_res16_stats) ----
at::Tensor row_stats_f16(const at::Tensor& x) {
  need(x, "x");
  TORCH_CHECK(x.dim() == 2 && x.scalar_type() == at::kHalf, "row_stats_f16: x fp16 [rows, C]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  at::Tensor st = at::empty({x.size(0), 2}, x.options().dtype(at::kFloat));
  const int rc = ovis_row_stats_f16(x.data_ptr(), st.data_ptr<float>(), x.size(0), (int)x.size(1), 1e-5f, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "row_stats_f16: ", ovis_last_error());
  return st;
}
at::Tensor row_stats_f16_meta(const at::Tensor& x) { return at::empty_symint({x.sym_size(0), 2}, x.options().dtype(at::kFloat)); }

at::Tensor gemm_nt_f16_ln(const at::Tensor& x, const at::Tensor& wg, const at::Tensor& s, const at::Tensor& c, const at::Tensor& stats, int64_t act) {
  need(x, "x"); need(wg, "wg"); need(s, "s"); need(c, "c"); need(stats, "stats");
  TORCH_CHECK(x.dim() == 2 && wg.dim() == 2 && x.size(1) == wg.size(1) && x.scalar_type() == at::kHalf && wg.scalar_type() == at::kHalf,
              "gemm_nt_f16_ln: x fp16 [M,K], wg fp16 [N,K]");
  const int M = (int)x.size(0), K = (int)x.size(1), N = (int)wg.size(0);
  TORCH_CHECK(s.scalar_type() == at::kFloat && c.scalar_type() == at::kFloat && stats.scalar_type() == at::kFloat && s.numel() == N &&
              c.numel() == N && stats.numel() == 2ll * M, "gemm_nt_f16_ln: s, c f32 [N]; stats f32 [M,2]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
  at::Tensor out = at::empty({M, N}, x.options());
  const int rc = ovis_gemm_nt_f16_ln(x.data_ptr(), K, wg.data_ptr(), K, out.data_ptr(), N, M, N, K, c.data_ptr<float>(), s.data_ptr<float>(),
                                     stats.data_ptr<float>(), (int)act, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "gemm_nt_f16_ln: ", ovis_last_error());
  return out;
}
at::Tensor gemm_nt_f16_ln_meta(const at::Tensor& x, const at::Tensor& wg, const at::Tensor&, const at::Tensor&, const at::Tensor&, int64_t) {
  return at::empty_symint({x.sym_size(0), wg.sym_size(0)}, x.options());
}

std::tuple<at::Tensor, at::Tensor> gemm_nt_f16_res16_stats(const at::Tensor& a, const at::Tensor& w, const c10::optional<at::Tensor>& bias,
                                                           const at::Tensor& residual) {
  need(a, "a"); need(w, "w"); need(residual, "residual");
  TORCH_CHECK(a.dim() == 2 && w.dim() == 2 && a.size(1) == w.size(1) && a.scalar_type() == at::kHalf && w.scalar_type() == at::kHalf &&
              residual.scalar_type() == at::kHalf, "gemm_nt_f16_res16_stats: fp16 a [M,K], w [N,K], residual [M,N]");
  const int M = (int)a.size(0), K = (int)a.size(1), N = (int)w.size(0);
  TORCH_CHECK(residual.dim() == 2 && residual.size(0) == M && residual.size(1) == N && N % 256 == 0, "gemm_nt_f16_res16_stats: residual [M,N], N % 256 == 0");
  if (bias.has_value() && bias->defined()) { need(*bias, "bias"); TORCH_CHECK(bias->scalar_type() == at::kFloat && bias->numel() == N, "gemm_nt_f16_res16_stats: bias f32 [N]"); }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(a.device());
  const int slots = 4 * (N / 256);
  at::Tensor out = at::empty({M, N}, a.options());
  at::Tensor part = at::empty({M, slots, 2}, a.options().dtype(at::kFloat));
  const int rc = ovis_gemm_nt_f16_res16_stats(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, fptr(bias), residual.data_ptr(), N,
                                              part.data_ptr<float>(), cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "gemm_nt_f16_res16_stats: ", ovis_last_error());
  return {out, part};
}
std::tuple<at::Tensor, at::Tensor> gemm_nt_f16_res16_stats_meta(const at::Tensor& a, const at::Tensor& w, const c10::optional<at::Tensor>&,
                                                                const at::Tensor&) {
  return {at::empty_symint({a.sym_size(0), w.sym_size(0)}, a.options()),
          at::empty_symint({a.sym_size(0), c10::SymInt(4) * (w.sym_size(0) / 256), c10::SymInt(2)}, a.options().dtype(at::kFloat))};
}
// partial (sum, sum of squares) pairs [M, slots, 2] of rows of C values -> (mean, rstd) [M, 2]
at::Tensor row_stats_finalize(const at::Tensor& part, int64_t C) {
  need(part, "part");
  TORCH_CHECK(part.dim() == 3 && part.size(2) == 2 && part.scalar_type() == at::kFloat, "row_stats_finalize: part f32 [M, slots, 2]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(part.device());
  at::Tensor stats = at::empty({part.size(0), 2}, part.options());
  const int rc = ovis_row_stats_finalize(part.data_ptr<float>(), (int)part.size(1), stats.data_ptr<float>(), part.size(0), (int)C, 1e-5f, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "row_stats_finalize: ", ovis_last_error());
  return stats;
}
at::Tensor row_stats_finalize_meta(const at::Tensor& part, int64_t) { return at::empty_symint({part.sym_size(0), 2}, part.options()); }

// ---- msda_encoder_fused --------------------------------------------------------------------------------------------
at::Tensor msda_encoder_fused(const at::Tensor& value, const at::Tensor& oa, const at::Tensor& shapes, const at::Tensor& lsi,
                              int64_t M, int64_t L, int64_t P) {
  need(value, "value"); need(oa, "oa"); need(shapes, "spatial_shapes"); need(lsi, "level_start_index");
  TORCH_CHECK(value.dim() == 3 && oa.dim() == 3 && value.scalar_type() == at::kFloat && oa.scalar_type() == at::kFloat,
              "msda_encoder_fused: value [B,S,C] f32, oa [B,S,M*L*P*3] f32");
  TORCH_CHECK(shapes.scalar_type() == at::kLong && lsi.scalar_type() == at::kLong, "msda_encoder_fused: int64 shapes / level_start_index");
  const int B = (int)value.size(0), S = (int)value.size(1), C = (int)value.size(2);
  TORCH_CHECK(C % M == 0 && oa.size(0) == B && oa.size(1) == S, "msda_encoder_fused: shape mismatch");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
  at::Tensor out = at::empty_like(value);
  const int rc = ovis_msda_encoder_fused_f32(value.data_ptr<float>(), oa.data_ptr<float>(), (int)oa.size(2), shapes.data_ptr<int64_t>(),
                                             lsi.data_ptr<int64_t>(), out.data_ptr<float>(), B, S, (int)M, C / (int)M, (int)L, (int)P, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "msda_encoder_fused: ", ovis_last_error());
  return out;
}
at::Tensor msda_encoder_fused_meta(const at::Tensor& value, const at::Tensor&, const at::Tensor&, const at::Tensor&, int64_t, int64_t, int64_t) {
  return at::empty_like(value);
}

// ---- attention_f16 -------------------------------------------------------------------------------------------------
// q / k / v are VIEWS into fp16 buffers: element (b, row, h, d) sits at data_ptr + (b*bs + row*ld + h*D + d) halfs
at::Tensor attention_f16(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                         int64_t q_bs, int64_t q_ld, int64_t k_bs, int64_t k_ld, int64_t v_bs, int64_t v_ld) {
  for (const at::Tensor* t : {&q, &k, &v}) TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kHalf, "attention_f16 needs fp16 HIP tensors");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
  at::Tensor out = at::empty({B, Nq, H * D}, q.options());
  const int rc = ovis_attention_f16(q.data_ptr(), q_bs, (int)q_ld, k.data_ptr(), k_bs, (int)k_ld, v.data_ptr(), v_bs, (int)v_ld, out.data_ptr(),
                                    Nq * H * D, (int)(H * D), (int)B, (int)H, (int)Nq, (int)Nk, (int)D, 1.0f / std::sqrt((float)D), cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "attention_f16: ", ovis_last_error());
  return out;
}
at::Tensor attention_f16_meta(const at::Tensor& q, const at::Tensor&, const at::Tensor&, int64_t B, int64_t H, int64_t Nq, int64_t, int64_t D,
                              int64_t, int64_t, int64_t, int64_t, int64_t, int64_t) {
  return at::empty({B, Nq, H * D}, q.options());
}

// ---- mask_bbox / clip_crop_patches ---------------------------------------------------------------------------------
at::Tensor mask_bbox(const at::Tensor& masks, int64_t Hp, int64_t Wp) {
  need(masks, "masks");
  TORCH_CHECK(masks.dim() == 4 && masks.scalar_type() == at::kFloat, "mask_bbox: masks [Q,T,h,w] f32 logits");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(masks.device());
  at::Tensor boxes = at::empty({masks.size(1), masks.size(0), 4}, masks.options().dtype(at::kInt));
  const int rc = ovis_mask_bbox(masks.data_ptr<float>(), boxes.data_ptr<int>(), (int)masks.size(0), (int)masks.size(1), (int)masks.size(2),
                                (int)masks.size(3), (int)Hp, (int)Wp, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "mask_bbox: ", ovis_last_error());
  return boxes;
}
at::Tensor mask_bbox_meta(const at::Tensor& masks, int64_t, int64_t) {
  return at::empty_symint({masks.sym_size(1), masks.sym_size(0), 4}, masks.options().dtype(at::kInt));
}

int64_t patch_row_len(int64_t patch) { return (3 * patch * patch + 7) / 8 * 8; }   // 16-byte fp16 rows (ViT-L/14: 588 -> 592)

at::Tensor clip_crop_patches(const at::Tensor& frames, const at::Tensor& masks, const at::Tensor& crops, int64_t Hp, int64_t Wp,
                             int64_t resolution, int64_t patch, at::ArrayRef<double> mean, at::ArrayRef<double> std_, bool out_f16) {
  need(frames, "frames"); need(masks, "masks"); need(crops, "crops");
  TORCH_CHECK(frames.scalar_type() == at::kByte && frames.dim() == 4 && masks.scalar_type() == at::kFloat && masks.dim() == 4 &&
                  crops.scalar_type() == at::kInt && crops.dim() == 2 && crops.size(1) == 6 && mean.size() == 3 && std_.size() == 3,
              "clip_crop_patches: frames uint8 [T,3,H,W], masks f32 [Q,T,h,w], crops int32 [M,6], mean / std of 3");
  const int64_t M = crops.size(0), G = resolution / patch, ld = patch_row_len(patch);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(frames.device());
  const auto opt = masks.options().dtype(out_f16 ? at::kHalf : at::kFloat);
  at::Tensor A = ld == 3 * patch * patch ? at::empty({M * G * G, ld}, opt) : at::zeros({M * G * G, ld}, opt);   // pad columns stay zero
  const float m3[3] = {(float)mean[0], (float)mean[1], (float)mean[2]}, s3[3] = {(float)std_[0], (float)std_[1], (float)std_[2]};
  // workspace of the leader / follower passes (crops of a frame that share a box compute the frame half once): from torch's caching
  // allocator, so that forwards on different streams never share it
  const int64_t wsb = ovis_clip_crop_workspace_bytes((int)M, (int)resolution);
  at::Tensor ws = at::empty({(wsb + 3) / 4}, masks.options());
  const int rc = ovis_clip_crop_patches_ws(frames.data_ptr<uint8_t>(), masks.data_ptr<float>(), crops.data_ptr<int>(), A.data_ptr(), nullptr, out_f16 ? 1 : 0,
                                           (int)M, (int)masks.size(0), (int)frames.size(0), (int)frames.size(2), (int)frames.size(3),
                                           (int)masks.size(2), (int)masks.size(3), (int)Hp, (int)Wp, (int)resolution, (int)patch, ld, m3, s3,
                                           ws.data_ptr(), wsb, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "clip_crop_patches: ", ovis_last_error());
  return A;
}
at::Tensor clip_crop_patches_meta(const at::Tensor&, const at::Tensor& masks, const at::Tensor& crops, int64_t, int64_t, int64_t resolution,
                                  int64_t patch, at::ArrayRef<double>, at::ArrayRef<double>, bool out_f16) {
  const int64_t G = resolution / patch;
  return at::empty_symint({crops.sym_size(0) * G * G, c10::SymInt(patch_row_len(patch))}, masks.options().dtype(out_f16 ? at::kHalf : at::kFloat));
}

// ---- hungarian_link / topk_entropy ---------------------------------------------------------------------------------
at::Tensor hungarian_link(const at::Tensor& embeds) {
  need(embeds, "embeds");
  TORCH_CHECK(embeds.dim() == 3 && embeds.scalar_type() == at::kFloat, "hungarian_link: embeds f32 [T,Q,C]");
  const int T = (int)embeds.size(0), Q = (int)embeds.size(1), C = (int)embeds.size(2);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(embeds.device());
  at::Tensor idx = at::empty({T, Q}, embeds.options().dtype(at::kInt));
  at::Tensor ws = at::empty({(int64_t)(ovis_hungarian_link_workspace_bytes(T, Q, C) / 4)}, embeds.options());
  const int rc = ovis_hungarian_link_f32(embeds.data_ptr<float>(), idx.data_ptr<int>(), ws.data_ptr<float>(), T, Q, C, cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "hungarian_link: ", ovis_last_error());
  return idx;
}
at::Tensor hungarian_link_meta(const at::Tensor& embeds) {
  return at::empty_symint({embeds.sym_size(0), embeds.sym_size(1)}, embeds.options().dtype(at::kInt));
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> topk_entropy(const at::Tensor& probs, const at::Tensor& row_ids, int64_t topk) {
  need(probs, "probs"); need(row_ids, "row_ids");
  TORCH_CHECK(probs.dim() == 2 && probs.scalar_type() == at::kFloat && row_ids.scalar_type() == at::kInt, "topk_entropy: probs f32 [rows,K], row_ids int32");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(probs.device());
  const auto oi = probs.options().dtype(at::kInt);
  at::Tensor idx = at::empty({topk}, oi), score = at::empty({topk}, probs.options()), ent = at::empty({topk}, probs.options()), sel = at::empty({topk}, oi);
  const int rc = ovis_topk_entropy_f32(probs.data_ptr<float>(), row_ids.data_ptr<int>(), (int)row_ids.numel(), (int)probs.size(1), (int)topk,
                                       idx.data_ptr<int>(), score.data_ptr<float>(), ent.data_ptr<float>(), sel.data_ptr<int>(), cur_stream());
  TORCH_CHECK(rc == OVIS_OK, "topk_entropy: ", ovis_last_error());
  return {idx, score, ent, sel};
}
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> topk_entropy_meta(const at::Tensor& probs, const at::Tensor&, int64_t topk) {
  const auto oi = probs.options().dtype(at::kInt);
  return {at::empty({topk}, oi), at::empty({topk}, probs.options()), at::empty({topk}, probs.options()), at::empty({topk}, oi)};
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(ovis_mi, m) {
  m.def("gemm_nt_f16(Tensor a, Tensor w, Tensor? bias, Tensor? residual, int act, bool out_f16) -> Tensor");
  m.def("gemm_nt_f16_ln(Tensor x, Tensor wg, Tensor s, Tensor c, Tensor stats, int act) -> Tensor");
  m.def("gemm_nt_f16_res16_stats(Tensor a, Tensor w, Tensor? bias, Tensor residual) -> (Tensor, Tensor)");
  m.def("row_stats_f16(Tensor x) -> Tensor");
  m.def("row_stats_finalize(Tensor part, int C) -> Tensor");
  m.def("msda_encoder_fused(Tensor value, Tensor offs_attn, Tensor spatial_shapes, Tensor level_start_index, int num_heads, int num_levels, "
        "int num_points) -> Tensor");
  m.def("attention_f16(Tensor q, Tensor k, Tensor v, int B, int H, int Nq, int Nk, int D, int q_bs, int q_ld, int k_bs, int k_ld, int v_bs, "
        "int v_ld) -> Tensor");
  m.def("mask_bbox(Tensor masks, int Hp, int Wp) -> Tensor");
  m.def("clip_crop_patches(Tensor frames, Tensor masks, Tensor crops, int Hp, int Wp, int resolution, int patch, float[] mean, float[] std, "
        "bool out_f16) -> Tensor");
  m.def("hungarian_link(Tensor embeds) -> Tensor");
  m.def("topk_entropy(Tensor probs, Tensor row_ids, int topk) -> (Tensor, Tensor, Tensor, Tensor)");
}
TORCH_LIBRARY_IMPL(ovis_mi, CUDA, m) {       // "CUDA" is the HIP device's dispatch key on ROCm builds of torch
  m.impl("gemm_nt_f16", &gemm_nt_f16);
  m.impl("gemm_nt_f16_ln", &gemm_nt_f16_ln);
  m.impl("gemm_nt_f16_res16_stats", &gemm_nt_f16_res16_stats);
  m.impl("row_stats_f16", &row_stats_f16);
  m.impl("row_stats_finalize", &row_stats_finalize);
  m.impl("msda_encoder_fused", &msda_encoder_fused);
  m.impl("attention_f16", &attention_f16);
  m.impl("mask_bbox", &mask_bbox);
  m.impl("clip_crop_patches", &clip_crop_patches);
  m.impl("hungarian_link", &hungarian_link);
  m.impl("topk_entropy", &topk_entropy);
}
TORCH_LIBRARY_IMPL(ovis_mi, Meta, m) {
  m.impl("gemm_nt_f16", &gemm_nt_f16_meta);
  m.impl("gemm_nt_f16_ln", &gemm_nt_f16_ln_meta);
  m.impl("gemm_nt_f16_res16_stats", &gemm_nt_f16_res16_stats_meta);
  m.impl("row_stats_f16", &row_stats_f16_meta);
  m.impl("row_stats_finalize", &row_stats_finalize_meta);
  m.impl("msda_encoder_fused", &msda_encoder_fused_meta);
  m.impl("attention_f16", &attention_f16_meta);
  m.impl("mask_bbox", &mask_bbox_meta);
  m.impl("clip_crop_patches", &clip_crop_patches_meta);
  m.impl("hungarian_link", &hungarian_link_meta);
  m.impl("topk_entropy", &topk_entropy_meta);
}
