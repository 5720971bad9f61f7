/* openvis_hip.h — C ABI of libopenvis_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-frame dense inference path of clownrat6/OpenVIS.
 * Every pointer is a DEVICE pointer unless its name ends in _host. Every entry
 * point enqueues on `stream` (a hipStream_t passed as void*; NULL = the null
 * stream), never synchronises, allocates nothing, keeps no global state and is
 * re-entrant. Return value: 0 on success, otherwise an OVIS_E* code; the message
 * is available from ovis_last_error() (thread-local). A failed call has launched
 * nothing. (The reference only printf()s launch errors — ms_deform_im2col_cuda.cuh:953-957.)
 *
 * Reference file:line citations are into /root/reference/.
 */
#ifndef OPENVIS_HIP_H
#define OPENVIS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVIS_OK 0
#define OVIS_EINVAL 1   /* bad argument (null pointer, non-positive size, unsupported shape) */
#define OVIS_ELAUNCH 2  /* hipLaunchKernel / runtime error */

typedef void* ovis_stream_t; /* hipStream_t */

/* Library ABI version (bumped when a signature changes). */
int ovis_abi_version(void);
/* Last error message of the calling thread ("" if none). */
const char* ovis_last_error(void);

/* ---- B1: multi-scale deformable attention forward ("K1") -------------------------------
 * Replaces pybind `MultiScaleDeformableAttention.ms_deform_attn_forward`
 *   openvis/modeling/pixel_decoder/ops/src/vision.cpp:19, ms_deform_attn.h:26-44,
 *   cuda/ms_deform_attn_cuda.cu:25-85, kernel cuda/ms_deform_im2col_cuda.cuh:242-304.
 *   value            [batch, spatial_size, num_heads, channels]      contiguous
 *   spatial_shapes   [num_levels, 2] int64 (H_l, W_l)                 (device, as in the reference)
 *   level_start_index[num_levels]    int64
 *   sampling_loc     [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) in [0,1]
 *   attn_weight      [batch, num_query, num_heads, num_levels, num_point]
 *   out              [batch, num_query, num_heads*channels]           fully overwritten
 * The reference's im2col_step chunking (cuda.cu:55-80) is a launch detail with no
 * numerical effect; one launch covers the whole batch here. */
int ovis_msda_forward_f32(const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc,
                          const float* attn_weight, float* out, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query,
                          int num_point, ovis_stream_t stream);
int ovis_msda_forward_f64(const double* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const double* sampling_loc,
                          const double* attn_weight, double* out, int batch, int spatial_size,
                          int num_heads, int channels, int num_levels, int num_query,
                          int num_point, ovis_stream_t stream);

/* ---- Dense layers and convolutions on the f32 matrix cores ---------------------------------
 * Replace the cuBLAS/cuDNN work behind the reference's nn.Linear / Conv2d modules on the path, e.g.
 *   ops/modules/ms_deform_attn.py:98-104,124 (value_proj, sampling_offsets, attention_weights, output_proj),
 *   pixel_decoder/msdeformattn.py:118-121 (FFN), 227-235 (input_proj), 260-267, 276-296 (FPN convs),
 *   transformer_decoder/video_mask2former_transformer_decoder.py:175-179, 204-216 (FFN, MLP heads),
 *   detectron2 ResNet-50 bottlenecks (Base.yaml:2-16), mask_adapted_clip/model.py:238-268 (ViT blocks).
 * Activation codes: 0 none, 1 ReLU, 2 QuickGELU x*sigmoid(1.702x) (model.py:232-234).
 *
 * ovis_gemm_nt_f32:  C[m,n] = act( sum_k A[m,k]*B[n,k] + bias[n] + residual[m,n] )
 *   A [M,K] row stride lda, B [N,K] row stride ldb (a torch Linear weight as stored), C [M,N] row stride ldc;
 *   bias [N] or NULL; residual [M,N] row stride ldr or NULL. Exact-f32 MFMA, f32 accumulation. */
int ovis_gemm_nt_f32(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                     int M, int N, int K, const float* bias, const float* residual, long long ldr, int act,
                     ovis_stream_t stream);
/* ovis_conv2d_nhwc_f32: y[n,oh,ow,co] = act( conv(x, w) + bias[co] + residual[n,oh,ow,co] ), implicit GEMM.
 *   x [N,H,W,Cin] (Cin % 4 == 0), w [Cout,KH,KW,Cin] (the reference's [Cout,Cin,KH,KW] weight permuted once
 *   at load), y/residual [N,OH,OW,Cout]; square stride/zero padding as nn.Conv2d. */
int ovis_conv2d_nhwc_f32(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, const float* bias, const float* residual,
                         int act, ovis_stream_t stream);

/* ---- HBM-bound helpers -----------------------------------------------------------------------
 * A1 pre-processing: out[t,y,x,0:3] = (frames[t,c,y,x] - mean[c]) / std[c], zero outside HxW and in channel 3.
 *   Replaces openvis/openvis.py:57-62 ((x - pixel_mean)/pixel_std + ImageList.from_tensors(.., 32)).
 *   frames uint8 [T,3,H,W]; out f32 [T,Hp,Wp,4] (NHWC, 4th channel zero); mean/std: 3 floats each (HOST). */
int ovis_preprocess_u8_nhwc4(const uint8_t* frames, float* out, int T, int H, int W, int Hp, int Wp,
                             const float* mean3_host, const float* std3_host, ovis_stream_t stream);
/* 3x3 / stride 2 / pad 1 max-pool, NHWC (detectron2 BasicStem, Base.yaml:2-16). y [N,OH,OW,C]. */
int ovis_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, ovis_stream_t stream);
/* y = LayerNorm(x + residual) * gamma + beta over the last dim (residual may be NULL), biased variance.
 *   msdeformattn.py:139-141,118-122 (norm1/norm2), video decoder:57-60,117-120,175-179, model.py:223-229. */
int ovis_layernorm_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                       long long rows, int C, float eps, ovis_stream_t stream);
/* GroupNorm(G) on NHWC + optional ReLU + optional "+ bilinear-resized(up_add)" (FPN top-down add):
 *   msdeformattn.py:227-235 (input_proj GN), 276-296 + 369-372 (lateral GN + F.interpolate add, output GN + ReLU).
 *   stats_ws: workspace of 2*G*N doubles. up_add [N,UH,UW,C] or NULL. */
int ovis_groupnorm_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta, double* stats_ws,
                            int N, int H, int W, int C, int G, float eps, int relu, const float* up_add, int UH,
                            int UW, ovis_stream_t stream);
/* out[i] = a[i] + b[i % nb]  (src + pos with pos shared by all frames; msdeformattn.py:138 with_pos_embed). */
int ovis_add_bcast_f32(const float* a, const float* b, float* out, long long n, long long nb, ovis_stream_t stream);
/* Sine position encodings, channel-last: 2-D (pixel_decoder/position_encoding.py:29-53) -> out [H,W,2*npf] with T=1,
 * 3-D (transformer_decoder/position_encoding.py:135-165) -> out [T,H,W,2*npf]; add_c [2*npf] or NULL is added per
 * channel (level_embed, msdeformattn.py:90). */
int ovis_pe_sine_f32(float* out, int T, int H, int W, int num_pos_feats, int three_d, const float* add_c,
                     ovis_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
