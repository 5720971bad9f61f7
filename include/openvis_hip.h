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

/* 16-bit VALUE storage (SURVEY.md 8(b) "+ bf16-value variant"): value is bf16 (or fp16) [batch, spatial_size, num_heads, channels]
 * (channels % 4 == 0), sampling_loc / attn_weight / out f32.  The taps are widened exactly and the arithmetic is that of
 * ovis_msda_forward_f32: the result equals the f32 op on the widened value tensor BIT FOR BIT, while a tap moves half the bytes.
 * The model path keeps f32 values (the reference's pixel decoder is f32, msdeformattn.py:329); these entry points serve callers
 * whose value projection is already 16-bit. */
int ovis_msda_forward_bf16v(const void* value_bf16, const int64_t* spatial_shapes, const int64_t* level_start_index,
                            const float* sampling_loc, const float* attn_weight, float* out, int batch, int spatial_size,
                            int num_heads, int channels, int num_levels, int num_query, int num_point, ovis_stream_t stream);
int ovis_msda_forward_f16v(const void* value_f16, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* sampling_loc, const float* attn_weight, float* out, int batch, int spatial_size,
                           int num_heads, int channels, int num_levels, int num_query, int num_point, ovis_stream_t stream);

/* Arithmetic used by the f32 GEMM / convolution entry points for large problems (>= 256 tiles of 128x128):
 *   1 (default)  bf16x3: every f32 operand is split exactly into three bf16 values while staged into LDS and six bf16
 *                MFMA products are accumulated in f32 -- same accuracy class as an f32 fmaf chain, ~2.5x the rate of
 *   0            the native v_mfma_f32_32x32x2_f32 kernel (an exact f32 fmaf chain), which small problems always use.
 *   2            bf16x2: the two leading planes, three products (16 significand bits per operand); 3 = what openvis_amd sets for its
 *                "fp16x2" policy: these entry points behave as under 1, the constant-weight layers go through the *_h2 entry points.
 * Replaces nothing in the reference (torch picks cuBLAS algorithms implicitly); process-wide, not thread-safe. */
int ovis_set_f32_gemm_mode(int mode);

/* Constant f32 weights can be split ONCE into the three bf16 planes of the bf16x3 arithmetic (planes bf16 [3][n], x ==
 * p0 + p1 + p2 exactly); the *_w3 entry points then skip the per-tile split of the B operand (same results bit for bit).
 * They also take the f32 weights, which small problems / mode 0 keep using. */
int ovis_split_f32_to_bf16x3(const float* x, void* planes, long long n, ovis_stream_t stream);
int ovis_gemm_nt_f32_w3(const float* A, long long lda, const float* B, long long ldb, const void* W3, long long plane, float* C,
                        long long ldc, int M, int N, int K, const float* bias, const float* residual, long long ldr, int act,
                        ovis_stream_t stream);
int ovis_conv2d_nhwc_f32_w3(const float* x, const float* w, const void* w3, long long plane, float* y, int N, int H, int W, int Cin,
                            int Cout, int KH, int KW, int stride, int pad, const float* bias, const float* residual, int act,
                            ovis_stream_t stream);

/* "fp16x2" (round 4): f32-grade large GEMMs / convolutions at the MFMA cost of bf16x2.  Every f32 operand is carried as fp16 hi + fp16 lo
 * (hi = fp16(x), lo = fp16(x - hi): 11 + 11 significand bits) and the three products hi hi + hi lo + lo hi run on v_mfma_f32_*_f16 with f32
 * accumulation: every term down to 2^-22 |a b| WHERE lo is a normal fp16 number; an activation below ~1e-3 (a_scale 16) keeps an ABSOLUTE
 * error of 2^-29 instead (fp16 subnormal spacing / a_scale), i.e. the guarantee is f32-grade against the row's scale |A||W| + |b| + |R|, not a
 * relative 2^-22 per element.  fp16's narrow range is met by power-of-two scales (exact): constant weights are split ONCE
 * into two fp16 planes of w * w_scale (ovis_split_f32_to_f16x2; callers pick w_scale = 2^k with max |w| w_scale in [2^14, 2^15)), activations are
 * multiplied by a_scale while they are split in registers (ovis_set_f16x2, default 16: |a| < 4 094, absolute floor 2^-29), the accumulators
 * carry a_scale * w_scale and are scaled back before bias-free epilogue work.  An activation beyond the range makes the result non-finite; the
 * kernels then set *range_flag = 1 (device int, checked by the caller when it next synchronises: openvis_amd repeats the clip under bf16x3).
 * The *_h2 entry points mirror the *_w3 ones (same shapes, same fall-back to the exact f32 kernels for small problems) and do not depend on
 * ovis_set_f32_gemm_mode.  ovis_set_f16x2 is per host thread.  Replaces nothing in the reference (torch picks cuBLAS algorithms implicitly);
 * the layers are those listed under "Dense layers" below (msdeformattn.py:329 keeps them in f32). */
int ovis_set_f16x2(float a_scale, int* range_flag);
int ovis_split_f32_to_f16x2(const float* x, void* planes, long long n, float scale, ovis_stream_t stream);
int ovis_gemm_nt_f32_h2(const float* A, long long lda, const float* B, long long ldb, const void* H2, long long plane, float w_scale,
                        float* C, long long ldc, int M, int N, int K, const float* bias, const float* residual, long long ldr, int act,
                        ovis_stream_t stream);
int ovis_conv2d_nhwc_f32_h2(const float* x, const float* w, const void* h2, long long plane, float w_scale, float* y, int N, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, const float* bias, const float* residual, int act,
                            ovis_stream_t stream);
int ovis_gemm_nt_f32_h2_ln_eligible(const float* A, long long lda, const void* H2, long long ldb, long long plane, const float* C, long long ldc,
                                    int M, int N, int K, const float* bias, const float* residual, long long ldr);
int ovis_gemm_nt_f32_h2_ln(const float* A, long long lda, const void* H2, long long ldb, long long plane, float w_scale, float* C, long long ldc,
                           int M, int N, int K, const float* bias, const float* residual, long long ldr, const float* gamma,
                           const float* beta, float eps, ovis_stream_t stream);
/* One fp16x2 GEMM with two outputs and a row-periodic term on the second (round 4): W [N, K] = [W1 ; W2] (H2 = its ovis_split_f32_to_f16x2
 * planes), C1[m, n] = sum_k A[m,k] W[n,k] + bias[n] for n < col0 and C2[m, n - col0] = sum_k A[m,k] W[n,k] + bias[n] + R[m % r_rows, n - col0]
 * for n >= col0.  col0 a multiple of 256, N - col0 >= 256, r_rows >= 256; _eligible tells whether the ping-pong kernel takes the problem
 * (otherwise run two GEMMs).  Replaces value_proj and the sampling_offsets / attention_weights projections of a deformable-attention
 * encoder layer, whose inputs differ only by the frame-independent position embedding (/root/reference/openvis/modeling/pixel_decoder/ops/
 * modules/ms_deform_attn.py:98-104 called with query = with_pos_embed(src, pos), msdeformattn.py:138): R = pos W2^T, once per shape. */
int ovis_gemm_nt_f32_h2_dual_eligible(const float* A, long long lda, const void* H2, long long ldb, long long plane, const float* C1,
                                      long long ldc1, const float* C2, long long ldc2, int M, int N, int K, const float* bias, const float* R,
                                      long long ldr, int r_rows, int col0);
int ovis_gemm_nt_f32_h2_dual(const float* A, long long lda, const void* H2, long long ldb, long long plane, float w_scale, float* C1,
                             long long ldc1, float* C2, long long ldc2, int M, int N, int K, const float* bias, const float* R, long long ldr,
                             int r_rows, int col0, ovis_stream_t stream);
int ovis_conv3x3_padded_f32_h2_eligible(const float* xpad, const void* h2, long long plane, const float* y, int T, int H, int W, int Cin,
                                        int Cout, const float* bias, int act);
int ovis_conv3x3_padded_f32_h2(const float* xpad, const void* h2, long long plane, float w_scale, float* y, int T, int H, int W, int Cin,
                               int Cout, const float* bias, int act, ovis_stream_t stream);
const char* ovis_gemm_nt_f32_h2_kernel(const float* A, long long lda, const void* H2, long long ldb, long long plane, const float* C,
                                       long long ldc, int M, int N, int K, const float* bias, const float* residual, long long ldr,
                                       int act);

/* f32-grade GEMM from PRE-SPLIT operands (gemm_f16_pp.hip, X3 mode): A3 [3][M][lda] and W3 [3][N][ldb] are the exact 3-way bf16
 * splits of the f32 operands (plane strides in elements; ovis_split_f32_to_bf16x3[_v8], or a producer that writes planes), the
 * ping-pong kernel runs the six plane-pair products as one K axis of 6 K.  C: f32 [M,ldc] (out_planes 0; bias / residual / ReLU
 * as ovis_gemm_nt_f32) or, out_planes 1, the three bf16 planes of the result (no residual) -- the operand format of the next
 * GEMM, so a chain of linear layers never materialises its f32 intermediate (the FFN of msdeformattn.py:118-121).
 * Shapes: ovis_gemm_x3pp_eligible(M, N, K, has_bias) != 0 (>= 256 tiles of 256x256, K % 64 == 0, N % 8 == 0). */
int ovis_gemm_x3pp_eligible(int M, int N, int K, int has_bias);
int ovis_split_f32_to_bf16x3_v8(const float* x, void* planes, long long n, ovis_stream_t stream);
int ovis_gemm_nt_bf16x3_planes(const void* A3, long long lda, long long planeA, const void* W3, long long ldb, long long planeB,
                               void* C, long long ldc, long long planeC, int M, int N, int K, const float* bias,
                               const float* residual, long long ldr, int act, int out_planes, ovis_stream_t stream);

/* ---- Dense layers and convolutions on the f32 matrix cores ---------------------------------
 * Replace the cuBLAS/cuDNN work behind the reference's nn.Linear / Conv2d modules on the path, e.g.
 *   ops/modules/ms_deform_attn.py:98-104,124 (value_proj, sampling_offsets, attention_weights, output_proj),
 *   pixel_decoder/msdeformattn.py:118-121 (FFN), 227-235 (input_proj), 260-267, 276-296 (FPN convs),
 *   transformer_decoder/video_mask2former_transformer_decoder.py:175-179, 204-216 (FFN, MLP heads),
 *   detectron2 ResNet-50 bottlenecks (Base.yaml:2-16), mask_adapted_clip/model.py:238-268 (ViT blocks).
 * Activation codes: 0 none, 1 ReLU, 2 QuickGELU x*sigmoid(1.702x) (model.py:232-234), 3 GELU 0.5x(1+erf(x/sqrt 2))
 * (nn.GELU of the Swin MLP, backbone/swin.py:21-41).
 *
 * ovis_gemm_nt_f32:  C[m,n] = act( sum_k A[m,k]*B[n,k] + bias[n] + residual[m,n] )
 *   A [M,K] row stride lda, B [N,K] row stride ldb (a torch Linear weight as stored), C [M,N] row stride ldc;
 *   bias [N] or NULL; residual [M,N] row stride ldr or NULL. Exact-f32 MFMA, f32 accumulation. */
int ovis_gemm_nt_f32(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                     int M, int N, int K, const float* bias, const float* residual, long long ldr, int act,
                     ovis_stream_t stream);
/* Batched form (independent problems z = 0..batch-1: A + z*a_bs, B + z*b_bs, C + z*c_bs; no residual): the per-frame
 *   mask einsum "bqc,bchw->bqhw" of the frame decoders (frame_mask2former_transformer_decoder.py:139-154). */
int ovis_gemm_nt_f32_batched(const float* A, long long lda, long long a_bs, const float* B, long long ldb,
                             long long b_bs, float* C, long long ldc, long long c_bs, int batch, int M, int N, int K,
                             const float* bias, int act, ovis_stream_t stream);
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
/* dec = LayerNorm(x) * gamma + beta; out = W2 relu(W1 relu(W0 dec + b0) + b1) + b2 in ONE launch: decoder_norm and the mask-embedding MLP of
 * the masked-attention decoders' prediction heads (/root/reference/openvis/modeling/transformer_decoder/video_mask2former_transformer_decoder.py:
 * 454-458, MLP :204-216).  x [rows, C], C == 256; wt0..wt2 are the Linear weights TRANSPOSED ([in, out], row-major); dec may be NULL. */
int ovis_ln_mlp3_f32(const float* x, const float* gamma, const float* beta, const float* wt0, const float* b0, const float* wt1,
                     const float* b1, const float* wt2, const float* b2, float* dec, float* out, int rows, int C, float eps,
                     ovis_stream_t stream);
/* Same, output written as fp16 (operand of the fp16 CLIP GEMMs; statistics and affine in f32). */
int ovis_layernorm_f32_to_f16(const float* x, const float* residual, const float* gamma, const float* beta, void* y_f16,
                              long long rows, int C, float eps, ovis_stream_t stream);
/* GroupNorm(G) on NHWC + optional ReLU + optional "+ bilinear-resized(up_add)" (FPN top-down add):
 *   msdeformattn.py:227-235 (input_proj GN), 276-296 + 369-372 (lateral GN + F.interpolate add, output GN + ReLU).
 *   stats_ws: workspace of 2*G*N doubles. up_add [N,UH,UW,C] or NULL. */
int ovis_groupnorm_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta, double* stats_ws,
                            int N, int H, int W, int C, int G, float eps, int relu, const float* up_add, int UH,
                            int UW, ovis_stream_t stream);
/* ... written into a zero-padded map y_padded [N][H+2][W+2][C] (ring of zeros stored by the same kernel): the input of ovis_conv3x3_padded_f32_w3 */
int ovis_groupnorm_nhwc_f32_padded(const float* x, float* y_padded, const float* gamma, const float* beta, double* stats_ws,
                                   int N, int H, int W, int C, int G, float eps, int relu, const float* up_add, int UH,
                                   int UW, ovis_stream_t stream);
/* out[i] = a[i] + b[i % nb]  (src + pos with pos shared by all frames; msdeformattn.py:138 with_pos_embed). */
int ovis_add_bcast_f32(const float* a, const float* b, float* out, long long n, long long nb, ovis_stream_t stream);
/* Sine position encodings, channel-last: 2-D (pixel_decoder/position_encoding.py:29-53) -> out [H,W,2*npf] with T=1,
 * 3-D (transformer_decoder/position_encoding.py:135-165) -> out [T,H,W,2*npf]; add_c [2*npf] or NULL is added per
 * channel (level_embed, msdeformattn.py:90). */
int ovis_pe_sine_f32(float* out, int T, int H, int W, int num_pos_feats, int three_d, const float* add_c,
                     ovis_stream_t stream);

/* Kernel selection of ovis_gemm_nt_f16 for problems of >= 256 tiles of 256x256 (the CLIP ViT GEMMs; replaces nothing in
 * the reference, torch picks cuBLAS algorithms implicitly).  mode 1 (default): the ping-pong kernel (gemm_f16_pp.hip:
 * the two wavefronts of a SIMD run half a phase apart, LDS-DMA 4 half-tiles ahead, 16x16x32 MFMA); mode 0: the lock-step
 * 256x256 kernel of round 1 (gemm_f16.hip).  raster_group > 0: N tiles per column group of the tile raster;
 * desync_ns: start offsets of the 32 persistent workgroups of an XCD are spread over [0, desync_ns) (< 0: automatic, 24 us
 * for the residual GEMMs, 0 otherwise).  Process-wide, not thread-safe. */
int ovis_set_f16_gemm_mode(int mode, int raster_group, int desync_ns);
/* Name of the kernel ovis_gemm_nt_f16 launches for this problem (static string; for profiles and bench.py's roofline). */
const char* ovis_gemm_nt_f16_kernel(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K,
                                    const float* bias, const float* residual, long long ldr, int act, int out_f16);

/* ovis_gemm_nt_f32_w3 followed by LayerNorm over the N == 256 columns, in ONE kernel: C = LayerNorm(A W^T + b + R) * gamma + beta -- the
 * post-norm of the pixel decoder's encoder layers (msdeformattn.py:139-146: norm1 after output_proj + residual, norm2 after linear2 +
 * residual).  A 256-column tile of the ping-pong f32-A kernel holds whole rows, so its epilogue normalises them before they are stored
 * (statistics in f32, single pass, clamped variance).  Only where ovis_gemm_nt_f32_w3_ln_eligible != 0 (f32-GEMM mode 2 = bf16x2, a shape
 * the ping-pong kernel takes, N == 256, a residual); otherwise callers run ovis_gemm_nt_f32_w3 and ovis_layernorm_f32. */
int ovis_gemm_nt_f32_w3_ln_eligible(const float* A, long long lda, const void* W3, long long ldb, long long plane, const float* C, long long ldc,
                                    int M, int N, int K, const float* bias, const float* residual, long long ldr);
int ovis_gemm_nt_f32_w3_ln(const float* A, long long lda, const void* W3, long long ldb, long long plane, float* C, long long ldc, int M, int N,
                           int K, const float* bias, const float* residual, long long ldr, const float* gamma, const float* beta, float eps,
                           ovis_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolution (the FPN output convolutions, msdeformattn.py:287-296, 372) over an input that is ALREADY zero-padded:
 * xpad f32 [T][H+2][W+2][Cin] (ovis_groupnorm_nhwc_f32_padded writes it), w3 = the bf16 planes of w [Cout][3][3][Cin], y f32 [T][H][W][Cout].
 * The ping-pong f32-A kernel walks the padded image as a dense GEMM (M = T H W, K = 9 Cin): per-lane row bases, a wave-uniform tap offset per
 * K step, the LDS-DMA pipeline of the GEMM -- instead of the gathering im2col loader.  bf16x2 policy, Cin % 32 == 0, shapes the ping-pong
 * kernel takes (ovis_conv3x3_padded_f32_w3_eligible != 0); act 0 / 1 (ReLU); otherwise callers use ovis_conv2d_nhwc_f32_w3. */
int ovis_conv3x3_padded_f32_w3_eligible(const float* xpad, const void* w3, long long plane, const float* y, int T, int H, int W, int Cin,
                                        int Cout, const float* bias, int act);
int ovis_conv3x3_padded_f32_w3(const float* xpad, const void* w3, long long plane, float* y, int T, int H, int W, int Cin, int Cout,
                               const float* bias, int act, ovis_stream_t stream);

/* Name of the kernel ovis_gemm_nt_f32_w3 (and the 1x1 / stride 1 / pad 0 case of ovis_conv2d_nhwc_f32_w3) launches when it is the
 * ping-pong kernel's f32-A mode -- bf16x2 (mode 2) on shapes of >= 256 tiles of 256x256 with K % 32 == 0 and <= 35 % padded
 * columns --, "" when it is gemm_f32x3_kernel / gemm_f32_kernel (static string; for profiles and bench.py's roofline). */
const char* ovis_gemm_nt_f32_w3_kernel(const float* A, long long lda, const void* W3, long long ldb, long long plane, const float* C,
                                       long long ldc, int M, int N, int K, const float* bias, const float* residual, long long ldr,
                                       int act);

/* fp16 residual stream of the CLIP tower (what the reference's fp16 CLIP keeps between blocks, adapter.py:108-111):
 *   ovis_gemm_nt_f16_res16: C (fp16) = fp16( A B^T + bias + f32(R_f16) ), f32 accumulation, one rounding; only for the shapes the
 *     ping-pong kernel takes (ovis_gemm_nt_f16_res16_eligible != 0; callers route smaller problems through ovis_gemm_nt_f16);
 *   ovis_layernorm_f16_to_f16 / _to_f32: LayerNorm of fp16 rows with f32 statistics (model.py:157-163);
 *   ovis_vit_embed_ln_f16: ln_pre(cat(cls, fp16 patch embeddings) + pos) -> fp16 tokens (model.py:328-343);
 *   ovis_cast_f16_to_f32_rows: y[r, 0:C] (f32, dense) = x[r * ldx + 0:C] (fp16), e.g. the class-token rows. */
int ovis_gemm_nt_f16_res16_eligible(const void* C, const void* R16, long long lda, long long ldb, long long ldc, long long ldr,
                                    int M, int N, int K, const float* bias);
int ovis_gemm_nt_f16_res16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                           const float* bias, const void* R16, long long ldr, ovis_stream_t stream);
int ovis_layernorm_f16_to_f16(const void* x_f16, const float* gamma, const float* beta, void* y_f16, long long rows, int C, float eps,
                              ovis_stream_t stream);
int ovis_layernorm_f16_to_f32(const void* x_f16, const float* gamma, const float* beta, float* y, long long rows, int C, float eps,
                              ovis_stream_t stream);
int ovis_vit_embed_ln_f16(const void* patch_f16, const float* cls, const float* pos, const float* gamma, const float* beta,
                          void* out_f16, int M, int L1, int C, float eps, ovis_stream_t stream);
int ovis_cast_f16_to_f32_rows(const void* x_f16, long long ldx, float* y, long long rows, int C, ovis_stream_t stream);

/* LayerNorm folded into the GEMM that consumes it (ln_1 -> in_proj, ln_2 -> c_fc of a ResidualAttentionBlock, model.py:262-267, on
 * the fp16 residual stream): the normalised rows are never written.  With Wg = fp16(gamma * W) (folded once at load),
 * s[n] = sum_k f32(Wg[n,k]) and c[n] = b[n] + sum_k beta[k] W[n,k]:
 *   LayerNorm(x) W^T + b  =  rstd[m] * (x Wg^T - mean[m] s[n]) + c[n]
 *   ovis_row_stats_f16: stats[m] = (mean, 1/sqrt(var + eps)) of the fp16 row m, f32, the two-pass arithmetic of ovis_layernorm_f16_*;
 *   ovis_gemm_nt_f16_ln: C (fp16) = act(...) with f32 accumulation starting at -mean[m] s[n]; act 0 (none) or 2 (QuickGELU); only
 *     the shapes the ping-pong kernel takes (ovis_gemm_nt_f16_ln_eligible != 0: those of ovis_gemm_nt_f16's fp16-output path with
 *     2 N <= 6144 and <= 128 tiles per workgroup); otherwise callers run ovis_layernorm_f16_to_f16 + ovis_gemm_nt_f16. */
int ovis_row_stats_f16(const void* x_f16, float* stats, long long rows, int C, float eps, ovis_stream_t stream);
/*   ovis_gemm_nt_f16_res16_stats: ovis_gemm_nt_f16_res16 whose epilogue also writes part [M][4 N/256][2] = (sum, sum of squares) of
 *     every 64-column piece of the fp16 rows it stores (N % 256 == 0); ovis_row_stats_finalize: stats[m] = (mean, rstd) from those
 *     partial sums in a fixed order (var = E[x^2] - mean^2 in f32, clamped at 0) -- the residual stream is not read again. */
int ovis_gemm_nt_f16_res16_stats(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                                 const float* bias, const void* R16, long long ldr, float* part, ovis_stream_t stream);
int ovis_row_stats_finalize(const float* part, int slots, float* stats, long long rows, int C, float eps, ovis_stream_t stream);
int ovis_gemm_nt_f16_ln_eligible(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K, const float* c,
                                 const float* s_rows, const float* stats, int act);
int ovis_gemm_nt_f16_ln(const void* A, long long lda, const void* Wg, long long ldb, void* C, long long ldc, int M, int N, int K,
                        const float* c, const float* s_rows, const float* stats, int act, ovis_stream_t stream);

/* ovis_gemm_nt_f16: same contract with fp16 A [M,K] / B [N,K] (K, lda, ldb multiples of 8), f32 accumulation,
 *   f32 bias / residual, C written as f32 (out_f16 == 0) or fp16.  Used for the CLIP ViT GEMMs only — the reference
 *   runs CLIP in fp16 on the GPU (adapter.py:108-111; clip.load on cuda).  On tiles with 16-byte aligned rows the f32
 *   accumulators START at bias + residual (C = act((bias + residual) + sum_k a*b) in f32), so no epilogue load is
 *   exposed; C may alias residual (each tile reads its residual block before it writes that block). */
int ovis_gemm_nt_f16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N,
                     int K, const float* bias, const float* residual, long long ldr, int act, int out_f16,
                     ovis_stream_t stream);
/* "autocast" variants: f32 activations rounded to fp16 while staged into LDS, fp16 weights (cast once at load), f32
 *   accumulation / bias / residual / output.  The arithmetic of the reference's GPU path for the backbone convolutions
 *   and the decoder's Linear / einsum under torch.cuda.amp.autocast (train_net.py:241); the pixel decoder stays f32
 *   (msdeformattn.py:329).  K % 8 == 0 (conv: KH*KW*Cin % 8 == 0). */
int ovis_gemm_nt_f32a_f16w(const float* A, long long lda, const void* B16, long long ldb, float* C, long long ldc, int M,
                           int N, int K, const float* bias, const float* residual, long long ldr, int act,
                           ovis_stream_t stream);
int ovis_gemm_nt_f32a_f16w_batched(const float* A, long long lda, long long a_bs, const void* B16, long long ldb,
                                   long long b_bs, float* C, long long ldc, long long c_bs, int batch, int M, int N,
                                   int K, const float* bias, int act, ovis_stream_t stream);
int ovis_conv2d_nhwc_f32a_f16w(const float* x, const void* w16, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                               int KW, int stride, int pad, const float* bias, const float* residual, int act,
                               ovis_stream_t stream);
/* fp16 STORAGE between the convolutions of a ResNet bottleneck (round 4; detectron2 BottleneckBlock, configs/openvoc_ytvis/Base.yaml:2-16,
 * under torch.cuda.amp.autocast, train_net.py:241).  The tensors conv1 -> conv2 -> conv3 (and stem -> pool -> res2.0) have exactly one
 * kind of reader, the next convolution, which rounds them to fp16 while staging: writing them as fp16 leaves every MFMA operand
 * bit-identical and halves their bytes; the block outputs (residual stream) stay f32.
 *   ovis_gemm_nt_x16: ovis_gemm_nt_f32a_f16w with A f32 or fp16 (a_f16) and C f32 (+ f32 residual) or fp16 (c_f16, no residual);
 *   ovis_conv2d_nhwc_f32a_f16w_o16: ovis_conv2d_nhwc_f32a_f16w writing fp16 (the stem);
 *   ovis_maxpool3x3s2_nhwc_f16: the stem's pool on the fp16 map (max commutes with the rounding);
 *   ovis_conv_h16: 3x3 (pad 1) / 1x1 convolution, stride 1 / 2, of an fp16 NHWC map (Cin, Cout multiples of 64) with fp16 weights
 *     [Cout, k, k, Cin]: LDS-DMA operand ring (three K steps of 64 in flight), v_mfma_f32_32x32x16_f16, bias + ReLU (+ f32 residual when
 *     the output is f32) in the epilogue; y fp16 or f32 [T, OH, OW, Cout].  csrc/conv_h16.hip. */
int ovis_gemm_nt_x16(const void* A, int a_f16, long long lda, const void* B16, long long ldb, void* C, int c_f16, long long ldc, int M, int N,
                     int K, const float* bias, const float* residual, long long ldr, int act, ovis_stream_t stream);
/* conv3 + projection shortcut of a bottleneck as ONE GEMM over a concatenated K axis (round 4; detectron2 BottleneckBlock.forward:
 * out = conv3(out); out += shortcut(x); relu -- the shortcut tensor is neither written nor read back): y = act([A1 | A2] B^T + bias) with
 * B16 [N, K1 + K2] = [w3 | w_shortcut] (fp16) and bias = b3 + b_shortcut; fp16 operands, f32 accumulation, f32 output.  K1 % 64 == 0.
 *   ovis_gemm_nt_x16_2a: A1 [M, K1] and A2 [M, K2] dense fp16 (res2.0: conv2's output and the pooled stem output);
 *   ovis_conv1x1_pair_x16: A1 [T OH OW, K1] dense fp16, second source the f32 block input x2 [T, H, W, C2] at the pixels (s oy, s ox)
 *     (the stride-s 1x1 shortcut of res3.0 / res4.0 / res5.0), rounded to fp16 while staged. */
int ovis_gemm_nt_x16_2a(const void* A1_f16, long long lda1, int K1, const void* A2_f16, long long lda2, int K2, const void* B16, long long ldb,
                        float* C, long long ldc, int M, int N, const float* bias, int act, ovis_stream_t stream);
int ovis_conv1x1_pair_x16(const void* A1_f16, int K1, const float* x2, int T, int H, int W, int C2, int stride, const void* B16, float* y,
                          int N, const float* bias, int act, ovis_stream_t stream);
int ovis_conv2d_nhwc_f32a_f16w_o16(const float* x, const void* w16, void* y_f16, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                   int stride, int pad, const float* bias, int act, ovis_stream_t stream);
int ovis_maxpool3x3s2_nhwc_f16(const void* x, void* y, int N, int H, int W, int C, ovis_stream_t stream);
/* ... and stem + pool in ONE launch (round 4): conv 7x7 / stride 2 / pad 3 (kernel padded to 7 x 8 taps, FrozenBN folded) + ReLU -> fp16 ->
 * max pool 3x3 / stride 2 / pad 1 (detectron2 BasicStem).  x f32 [T, H, W, 4] (W even), w16 fp16 [64, 7, 8, 4], bias f32 [64] ->
 * y fp16 [T, PH, PW, 64].  Same values as ovis_conv2d_nhwc_f32a_f16w_o16 + ovis_maxpool3x3s2_nhwc_f16 up to the f32 summation order. */
int ovis_resnet_stem_pool_f16(const float* x, const void* w16, const float* bias, void* y_f16, int T, int H, int W, ovis_stream_t stream);
int ovis_conv_h16(const void* x_f16, const void* w_f16, void* y, int out_f16, int T, int H, int W, int Cin, int Cout, int ksize, int stride,
                  const float* bias, const float* residual, int act, ovis_stream_t stream);
/* y (fp16) = x (f32), n % 4 == 0 (weights are cast once at load). */
int ovis_cast_f32_to_f16(const float* x, void* y, long long n, ovis_stream_t stream);

/* ---- Multi-head attention (flash style, f32 MFMA) ----------------------------------------------
 * out[b,q,h*D:(h+1)*D] = softmax_k( scale * <Q[b,q,h], K[b,k,h]> , mask ) V[b,k,h]
 *   Replaces nn.MultiheadAttention's core in video decoder:52-62 (self), 110-122 + 417-426 (masked cross) and
 *   mask_adapted_clip/model.py:254-263 (ViT); in/out projections are ovis_gemm_nt_f32 calls.
 *   q/k/v/out: element (b, row, h, d) at ptr[b*bs + row*ld + h*D + d]; D in {32, 64}.
 *   mask: uint8 [Nq, mask_ld] (1 = blocked), shared by all heads, or NULL; mask_bs = 0: shared by all batches (the
 *   offline video decoder), otherwise batch b uses mask + b*mask_bs and row_open + b*Nq (per-frame decoders,
 *   frame_mask2former_transformer_decoder.py:85-94);
 *   bias: optional additive f32 bias added to the scaled scores before the softmax, element (b,h,q,k) at
 *   bias[b*bias_bs + h*bias_hs + q*bias_ld + k], rows padded to a multiple of 4 floats (the SideAdapter's attention
 *   bias, clip_adapter/side_adapter.py:70-78, 237-270), or NULL;
 *   row_open: int32 [Nq] = number of unblocked keys per row (rows with 0 are treated as unmasked,
 *   video decoder:419) or NULL.  nsplit > 1 splits the key range over workgroups (needs
 *   ovis_attention_workspace_bytes(B,H,Nq,D,nsplit) bytes of workspace).  out_f16 != 0 writes `out` as fp16
 *   (feeds the fp16 CLIP GEMMs; nsplit must be 1). */
long long ovis_attention_workspace_bytes(int B, int H, int Nq, int D, int nsplit);
int ovis_attention_f32(const float* q, long long q_bs, int q_ld, const float* k, long long k_bs, int k_ld,
                       const float* v, long long v_bs, int v_ld, void* out, long long o_bs, int o_ld, int out_f16,
                       const uint8_t* mask, long long mask_ld, long long mask_bs, const int* row_open, const float* bias,
                       long long bias_bs, long long bias_hs, int bias_ld, int B, int H, int Nq, int Nk, int D,
                       float scale, int nsplit, float* workspace, ovis_stream_t stream);

/* ---- One clip's masked cross-attention with the KEYS split over several GPUs (OpenVIS row: "split-KV") --------------
 *   The offline video decoder attends over the keys of all T frames at once (video decoder:397-403, 417-426).  With the frames of ONE clip
 *   sharded over GPUs every GPU holds the K / V rows of its own frames; per decoder layer it runs
 *     ovis_attention_partial_f32  -> this GPU's un-normalised flash partial, packed as
 *                                    [B*H*Nq*D weighted V | B*H*Nq*2 (row max in log2 units, row sum) | B*Nq "had an open key here"]
 *                                    = ovis_attention_partial_floats(B,H,Nq,D) floats (100 queries x 8 heads x 32: 109 200 B),
 *     one all-gather of that block over the GPUs (RCCL; the caller's job -- this library knows no communicator),
 *     ovis_attention_merge_f32    -> out[b,q,h*D:(h+1)*D] from the R gathered blocks (block r at parts + r*stride floats).
 *   mask / row_open cover THIS GPU's keys only.  A row that is blocked on every key of the clip attends to all keys (video decoder:419):
 *   the flash kernel runs locally-closed rows unmasked and the merge keeps those partials only if the row is closed on every GPU.
 *   R = 1 reproduces ovis_attention_f32 with the same nsplit (same partials, same merge arithmetic).
 *   workspace: ovis_attention_partial_workspace_bytes(B,H,Nq,D,nsplit) bytes (needed for every nsplit >= 1). */
long long ovis_attention_partial_floats(int B, int H, int Nq, int D);
long long ovis_attention_partial_workspace_bytes(int B, int H, int Nq, int D, int nsplit);
int ovis_attention_partial_f32(const float* q, long long q_bs, int q_ld, const float* k, long long k_bs, int k_ld,
                               const float* v, long long v_bs, int v_ld, const uint8_t* mask, long long mask_ld,
                               long long mask_bs, const int* row_open, int B, int H, int Nq, int Nk, int D, float scale,
                               int nsplit, float* workspace, float* partial, ovis_stream_t stream);
int ovis_attention_merge_f32(const float* parts, int R, long long stride, float* out, long long o_bs, int o_ld, int B, int H,
                             int Nq, int D, ovis_stream_t stream);

/* fp16-operand variant for the CLIP ViT tower (no mask, no split): q/k/v/out fp16, f32 softmax + accumulation, D = 64.
 *   element (b,row,h,d) at ptr[b*bs + row*ld + h*D + d] (strides in halfs, multiples of 8). */
int ovis_attention_f16(const void* q, long long q_bs, int q_ld, const void* k, long long k_bs, int k_ld, const void* v,
                       long long v_bs, int v_ld, void* out, long long o_bs, int o_ld, int B, int H, int Nq, int Nk,
                       int D, float scale, ovis_stream_t stream);

/* ---- OpenVIS-specific fused stages ---------------------------------------------------------------
 * Encoder deformable attention with the softmax over L*P and the sampling-location arithmetic fused in
 *   (ops/modules/ms_deform_attn.py:102-118 + msdeformattn.py:155-168 + cuh:242-304).
 *   offs_attn [B,S,ld_oa]: first M*L*P*2 columns = sampling_offsets output, next M*L*P = attention_weights
 *   logits (one fused GEMM); reference points are those of the encoder (valid ratios == 1). L=3, P=4. */
int ovis_msda_encoder_fused_f32(const float* value, const float* offs_attn, int ld_oa, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, float* out, int batch, int spatial_size,
                                int num_heads, int channels, int num_levels, int num_point, ovis_stream_t stream);
/* Same operation with the level shapes also given on the host (shapes_host int32 [3,2] = (H_l, W_l), coarse to fine): the
 * queries of the finest level run as 8x8 query tiles per head whose +/- `radius` pixel value windows are staged in LDS
 * (coalesced 128-byte row reads; taps beyond the radius fall back to a global load, so results are identical for any
 * offsets); the queries of the two coarser levels use the direct-gather kernel.  head_dim (channels) must be 32. */
int ovis_msda_encoder_fused_tiled_f32(const float* value, const float* offs_attn, int ld_oa, const int64_t* spatial_shapes,
                                      const int64_t* level_start_index, const int* shapes_host, float* out, int batch,
                                      int spatial_size, int num_heads, int channels, int num_levels, int num_point, int radius,
                                      ovis_stream_t stream);
/* mask[q,k] = sigmoid(logits[q,k]) < 0.5 ; row_open[q] = #unblocked (video decoder:465-469, 419). */
int ovis_attn_mask_from_logits(const float* logits, long long ld, uint8_t* mask, long long mask_ld, int* row_open,
                               int Q, int Nk, ovis_stream_t stream);
/* y = mean of the 2x2 centre taps of each s x s cell (== bilinear resize by exactly 1/s, align_corners False). */
int ovis_center_pool_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int s, ovis_stream_t stream);
/* boxes[t,q] = inclusive (x0,y0,x1,y1) of { sigmoid(bilinear_up(masks[q,t]) to Hp x Wp) > 0.5 }, x1 < 0 if empty
 *   (openvis.py:87-96,118; adapter.py:88-94; detectron2 BitMasks.get_bounding_boxes). masks [Q,T,h,w] logits. */
int ovis_mask_bbox(const float* masks, int* boxes, int Q, int T, int h, int w, int Hp, int Wp, ovis_stream_t stream);
/* Crop list built on the device (adapter.py:86-102 without the host read-back of the boxes): boxes int32 [T,Q,4] (ovis_mask_bbox) ->
 *   crops int32 [T*Q,6] = (t, q, x0, y0, x1, y1) for EVERY (frame, query) in that order (not compacted: no downstream shape depends on
 *   the data; an empty mask gets a box far outside the frame, for which ovis_clip_crop_patches writes the normalised zero image),
 *   slot int32 [T*Q] = t*Q + q or -1 (the aggregation's crop-row map), counts[0] += number of non-empty masks (caller zeroes it). */
int ovis_crop_list_static(const int* boxes, int* crops, int* slot, int* counts, int T, int Q, int Hp, int Wp, ovis_stream_t stream);
/* CLIP crops (adapter.py:96-116,140-143) written as the patch-embedding im2col matrix
 *   A[(m*G*G + py*G + px), c*ps*ps + iy*ps + ix]; crops int32 [M,6] = (t,q,x0,y0,x1,y1);
 *   frames uint8 [T,3,H,W] (raw, un-padded); masks [Q,T,h,w] logits; mean/std = CLIP's (HOST, in [0,1] units);
 *   A is f32, or fp16 when out_f16 != 0 (the reference feeds CLIP fp16 crops, adapter.py:108-111). */
int ovis_clip_crop_patches(const uint8_t* frames, const float* masks, const int* crops, void* A, int out_f16, int M,
                           int Q, int T, int H, int W, int h, int w, int Hp, int Wp, int resolution, int patch,
                           long long lda, const float* mean3_host, const float* std3_host, ovis_stream_t stream);
/* As ovis_clip_crop_patches, for AdaptedClipAdapter (mask_adapted_adapter.py:79-123, 143-147): additionally writes
 *   patch_open[m*G*G + py*G + px] = ceil(AvgPool2d(patch)(mask_region)) (model.py:332-333) as 0/1 bytes -- 1 iff any bin of
 *   the patch has a positive mask_region value.  patch_open (device, M*G*G bytes) is cleared by the call. */
int ovis_clip_crop_patches_masked(const uint8_t* frames, const float* masks, const int* crops, void* A,
                                  unsigned char* patch_open, int out_f16, int M, int Q, int T, int H, int W, int h, int w,
                                  int Hp, int Wp, int resolution, int patch, long long lda, const float* mean3_host,
                                  const float* std3_host, ovis_stream_t stream);
/* Both of the above with a caller-provided device workspace of >= ovis_clip_crop_workspace_bytes(M, resolution) bytes (16-byte aligned;
 *   patch_open may be NULL = ovis_clip_crop_patches).  The frame half of a crop -- roi_align of the RGB planes, adapter.py:104-108 -- depends
 *   on (frame, box) only: with the workspace, crops of one frame that share a box compute it once (a leader pass that keeps the bins' frame
 *   averages, a follower pass that evaluates the mask half only); bit-identical to the one-pass entry points, whatever the boxes are.
 *   ws == NULL: the one-pass kernel.  ovis_clip_crop_workspace_bytes returns 0 where the launcher would not use a workspace (M > 3 072: the
 *   dedupe kernel stages the crop list in LDS). */
long long ovis_clip_crop_workspace_bytes(int M, int resolution);
int ovis_clip_crop_patches_ws(const uint8_t* frames, const float* masks, const int* crops, void* A, unsigned char* patch_open, int out_f16,
                              int M, int Q, int T, int H, int W, int h, int w, int Hp, int Wp, int resolution, int patch, long long lda,
                              const float* mean3_host, const float* std3_host, void* ws, long long ws_bytes, ovis_stream_t stream);
/* Mask prompt of the mask-adapted CLIP ViT (third_parties/mask_adapted_clip/mask_adapted_clip/model.py:334-338 before
 *   the class token, :349-352 after block d < mask_prompt_depth): in place,
 *   x[m, first_token + l, :] = patch_open[m*L + l] ? x[m, first_token + l, :] : mask_embedding[emb_rows == 1 ? 0 : l, :];
 *   x f32 [M, tokens_per_item, C]; mask_embedding f32 [emb_rows, C] (one depth slice). */
int ovis_mask_prompt_select_f32(float* x, const unsigned char* patch_open, const float* mask_embedding, int M, int L, int C,
                                int tokens_per_item, int first_token, int emb_rows, ovis_stream_t stream);
/* ViT token assembly + ln_pre (model.py:341-343): out [M,L1,C]; patch [M,L1-1,C]; cls [C]; pos [L1,C]. */
int ovis_vit_embed_ln_f32(const float* patch, const float* cls, const float* pos, const float* gamma, const float* beta,
                          float* out, int M, int L1, int C, float eps, ovis_stream_t stream);
/* y[r,:] = x[r,:] / ||x[r,:]|| * scale (adapter.py:118-119,144,146). */
int ovis_l2norm_rows_f32(const float* x, float* y, long long rows, int C, float scale, ovis_stream_t stream);
/* probs[q,:] = softmax_K( mean over frames t with slot[t,q] >= 0 of crop_logits[slot[t,q],:] ) (openvis.py:130-141). */
int ovis_openvis_aggregate_f32(const float* crop_logits, const int* slot, float* probs, int* qvalid, int T, int Q, int K,
                               ovis_stream_t stream);
/* top-k over the flattened probabilities of rows row_ids (flat index = i*K + k over the compacted rows) + entropy
 *   of the selected rows (video_maskformer.py:267-272). */
int ovis_topk_entropy_f32(const float* probs, const int* row_ids, int nrows, int K, int topk, int* out_idx,
                          float* out_score, float* out_entropy, int* out_query /* row_ids[idx / K], may be NULL */,
                          ovis_stream_t stream);
/* final masks of the selected queries (openvis.py:87-96 + video_maskformer.py:273-278): out uint8 [n_sel,T,OH,OW], or
 * [n_sel,T,OW,OH] when column_major != 0 (the order in which COCO RLE scans a mask, see ovis_rle_encode_u8). */
int ovis_final_masks_u8(const float* masks, const int* sel_q, uint8_t* out, int n_sel, int Q, int T, int h, int w, int Hp,
                        int Wp, int H, int W, int OH, int OW, int column_major, ovis_stream_t stream);

/* ---- A13: SideAdapter (SAN / BriVIS) helpers -------------------------------------------------------
 * Front image path (side_adapter.py:150-153): F.interpolate(frames/255, (R,R), "bicubic") of the raw frames zero-padded
 *   from (H,W) to (Hp,Wp), CLIP mean/std normalisation, written as the patch-embedding im2col matrix
 *   A[(t*G*G + py*G + px), c*ps*ps + iy*ps + ix] (f32, or fp16 when out_f16 != 0). frames uint8 [T,3,H,W]. */
int ovis_san_front_patches(const uint8_t* frames, void* A, int out_f16, int T, int H, int W, int Hp, int Wp,
                           int resolution, int patch, long long lda, const float* mean3_host, const float* std3_host,
                           ovis_stream_t stream);
/* F.adaptive_max_pool2d over N planes [H,W] -> [OH,OW] (side_adapter.py:244, downsample2d(method="max")). */
int ovis_adaptive_maxpool2d_f32(const float* x, float* y, long long N, int H, int W, int OH, int OW, ovis_stream_t stream);
/* Additive attention bias of the back blocks (side_adapter.py:253-265): pooled [BN,Q,L] -> out [BN, Q+1+L, ld]. */
int ovis_san_attn_bias_f32(const float* pooled, float* out, long long BN, int Q, int L, int ld, ovis_stream_t stream);
/* dst [N,H,W,C] += bilinear_resize(src [N,h,w,C], (H,W)), align_corners False (msdeformattn.py:338-344). */
int ovis_bilinear_resize_add_nhwc_f32(float* dst, const float* src, int N, int H, int W, int C, int h, int w,
                                      ovis_stream_t stream);

/* ---- A14: temporal instance linker (MinVIS tracker) ----------------------------------------------
 * indices[t, i] = index of the frame-t query assigned to tracked slot i by the Hungarian chain of
 *   openvis/modeling/minvis.py:28-72 (cost 1 - cosine, scipy.optimize.linear_sum_assignment on target x current,
 *   targets of frame t = frame t-1's embeddings in their assigned order; frame 0 is matched against itself).
 *   embeds f32 [T,Q,C] (Q % 4 == 0, C % 4 == 0); indices int32 [T,Q]; workspace of
 *   ovis_hungarian_link_workspace_bytes(T,Q,C) bytes.  Row normalisation, one batched GEMM for every frame-to-frame
 *   cosine matrix, T INDEPENDENT single-wavefront Jonker-Volgenant solves (a row permutation of the cost matrix only
 *   permutes its optimal assignment, so frame t is solved on the un-permuted rows) and one composition of the T
 *   permutations (the reference does one GPU->CPU sync + scipy call per frame, sequentially). */
long long ovis_hungarian_link_workspace_bytes(int T, int Q, int C);
int ovis_hungarian_link_f32(const float* embeds, int* indices, float* workspace, int T, int Q, int C,
                            ovis_stream_t stream);
/* out[b, m, :] = src[b, idx[b, m], :] (openvis/utils/index.py:4-18 batch_index) with explicit strides (in floats):
 *   element row (b, n) of src starts at src + b*src_bs + n*src_rs; len floats per row (len % 4 == 0). */
int ovis_batch_index_rows_f32(const float* src, long long src_bs, long long src_rs, const int* idx, float* out,
                              long long out_bs, long long out_rs, int B, int M, long long len, ovis_stream_t stream);

/* ---- Swin backbone data movement (backbone/swin.py) --------------------------------------------------------------
 * x is a token map f32 [B,H,W,C] (C % 4 == 0); ws = window size, shift = cyclic shift (0 or ws/2);
 * Hp, Wp = H, W rounded up to multiples of ws; nW = (Hp/ws)*(Wp/ws).
 *
 * ovis_swin_window_partition_f32: swin.py:241-262 -- F.pad (zeros) + torch.roll(-shift) + window_partition:
 *   win[(b*nW + wy*(Wp/ws) + wx), iy*ws + ix, :] = xpad[b, (wy*ws+iy+shift) % Hp, (wx*ws+ix+shift) % Wp, :]
 * ovis_swin_window_merge_add_f32: swin.py:267-281 -- window_reverse + torch.roll(+shift) + crop + residual:
 *   out[b,y,x,:] = shortcut[b,y,x,:] + win[window/slot of ((y-shift) mod Hp, (x-shift) mod Wp)]
 * ovis_swin_shift_mask_u8: swin.py:381-404 -- mask[w, i, j] = 1 where the shifted-window attention mask is -100
 *   (tokens i and j of window w come from different regions), else 0; mask u8 [nW, ws*ws, ld] (ld >= ws*ws).
 * ovis_swin_patch_merge_gather_f32: swin.py:303-317 -- out[b, y, x, :] = cat(x[2y,2x], x[2y+1,2x], x[2y,2x+1],
 *   x[2y+1,2x+1]) with zeros beyond an odd H/W; out f32 [B, ceil(H/2), ceil(W/2), 4C].
 * ovis_swin_relpos_bias_f32: swin.py:147-155 -- bias[h, i, j] = table[index(i,j), h] with
 *   index = (yi-yj+ws-1)*(2ws-1) + (xi-xj+ws-1); table f32 [(2ws-1)^2, heads]; bias f32 [heads, ws*ws, ld]. */
/* ovis_swin_window_attention_f16: swin.py:130-169 (WindowAttention.forward) on fp16 operands with f32 softmax.
 *   qkv fp16 [nwin, N, 3C] (q | k | v, head h in columns h*32..h*32+31 of each third), N = ws*ws <= 160, head_dim 32;
 *   bias f32 [heads, N, ld] (ovis_swin_relpos_bias_f32), mask u8 [nW, N, ld] of the window (b mod nW) or NULL
 *   (ovis_swin_shift_mask_u8), out fp16 [nwin, N, C].  scale = head_dim ** -0.5. */
int ovis_swin_window_attention_f16(const void* qkv, void* out, const float* bias, const uint8_t* mask, long long nwin, int N,
                                   int C, int heads, int nW, int ld, float scale, ovis_stream_t stream);
int ovis_swin_window_partition_f32(const float* x, float* win, int B, int H, int W, int C, int ws, int shift,
                                   ovis_stream_t stream);
int ovis_swin_window_merge_add_f32(const float* win, const float* shortcut, float* out, int B, int H, int W, int C, int ws,
                                   int shift, ovis_stream_t stream);
int ovis_swin_shift_mask_u8(uint8_t* mask, int H, int W, int ws, int shift, int ld, ovis_stream_t stream);
int ovis_swin_patch_merge_gather_f32(const float* x, float* out, int B, int H, int W, int C, ovis_stream_t stream);
int ovis_swin_relpos_bias_f32(const float* table, float* bias, int heads, int ws, int ld, ovis_stream_t stream);

/* Prompt-ensemble mean (adapter.py:131-134: torch.stack(text_embeds_bucket).mean(dim=0)): y[i] = mean_t x[t*len + i]. */
int ovis_mean_dim0_f32(const float* x, float* y, int n, long long len, ovis_stream_t stream);

/* COCO run-length encoding on the GPU -- replaces the per-mask host loop of openvis/data/evals/ytvis_eval.py:258-301
 * (mask.cpu() -> np.array(order="F") -> pycocotools rleEncode).
 *   masks u8 [n_masks, len]: each mask flattened in COLUMN-major order (ovis_final_masks_u8 with column_major = 1);
 *   counts int32 [n_masks, cap]: uncompressed COCO counts (0-run first, may be 0); n_runs int32 [n_masks].
 *   n_runs[i] > cap means the buffer was too small for mask i (its first cap-1 counts are valid). */
int ovis_rle_encode_u8(const uint8_t* masks, int n_masks, long long len, int* counts, int* n_runs, int cap, ovis_stream_t stream);

/* Test-time input resize (augmentation.py:368-373 -> detectron2 ResizeShortestEdge -> PIL Image.resize(BILINEAR)),
 * bit-exact with Pillow's 22-bit fixed-point separable resampling.  src u8 [H,W,3] (decoded frame, device), tmp u8
 * [H,OW,3] scratch, dst u8 [3,OH,OW] (planar, what ovis_preprocess_u8_nhwc4 reads).  {x,y}bounds int32 [out,2] =
 * (first input index, tap count), {x,y}k int32 [out,ksize] fixed-point coefficients (openvis_amd/data.py builds them
 * as Pillow's precompute_coeffs + normalize_coeffs_8bpc). */
int ovis_pil_resize_u8_hwc_to_chw(const uint8_t* src, int H, int W, uint8_t* tmp, uint8_t* dst, int OH, int OW,
                                  const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize,
                                  ovis_stream_t stream);
/* The same resize whose vertical pass ALSO writes what ovis_preprocess_u8_nhwc4 would compute from dst (resize + normalise + pad
 * fused): img f32 [Hp,Wp,4] = ((dst - mean) / std, 0) zero padded (openvis.py:57-62), Hp >= OH, Wp >= OW; dst is still written (the CLIP crops
 * read the resized uint8 frame, adapter.py:96-108).  Both outputs bit-identical to the two separate calls. */
int ovis_pil_resize_preprocess_u8(const uint8_t* src, int H, int W, uint8_t* tmp, uint8_t* dst, float* img_nhwc4, int OH, int OW, int Hp,
                                  int Wp, const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize,
                                  const float* mean3_host, const float* std3_host, ovis_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
