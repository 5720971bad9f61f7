// f32-activation x fp16-weight GEMM / implicit-GEMM convolution with fp16 MFMA operands and f32 accumulation.
//
//   C[m,n] (f32) = act( sum_k fp16(A[m,k]) * B16[n,k] + bias[n] + R[m,n] )
//
// This is the "autocast" arithmetic of the reference's GPU path: under torch.cuda.amp.autocast (train_net.py:241)
// the backbone convolutions and the decoder's Linear / einsum run with fp16 operands and f32 accumulation, while the
// pixel decoder is forced to f32 (msdeformattn.py:329).  Activations stay f32 in HBM (no other kernel changes);
// they are rounded to fp16 while being staged into LDS, weights are cast once at load.
// Same tiling as gemm_f16.hip's register-staged kernel: 128x128x64 (or 64x64x64) tile, v_mfma_f32_32x32x16_f16,
// K-permuted fragments, rows padded to 144 B, operand roles swapped for the vectorised epilogue (gemm_epilogue.h).
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_loaders.h"
#include <type_traits>

namespace {

using ovis::ConvA;
using ovis::ConvGeom;
using ovis::DenseA;
using ovis::DenseH;
using ovis::DualA;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BKH = 64;
constexpr int LDS_ROW = BKH + 8;

__device__ __forceinline__ uint4 cvt8(bool ok0, float4 a, bool ok1, float4 b) {
  union { _Float16 h[8]; uint4 u; } o;
  o.h[0] = (_Float16)(ok0 ? a.x : 0.f); o.h[1] = (_Float16)(ok0 ? a.y : 0.f);
  o.h[2] = (_Float16)(ok0 ? a.z : 0.f); o.h[3] = (_Float16)(ok0 ? a.w : 0.f);
  o.h[4] = (_Float16)(ok1 ? b.x : 0.f); o.h[5] = (_Float16)(ok1 ? b.y : 0.f);
  o.h[6] = (_Float16)(ok1 ? b.z : 0.f); o.h[7] = (_Float16)(ok1 ? b.w : 0.f);
  return o.u;
}

// LoaderA = DenseH: A is already fp16 in memory (one 16-byte load per 8 k, no conversion).  OUT16: C is written as fp16 (no residual) --
// the tensors between the convolutions of a bottleneck (round 4): their only reader is the next convolution, which rounded the f32 copy
// to fp16 while staging, so the operands that reach the MFMA are bit-identical and the tensor costs half the bytes.
template <int BM, int BN, typename LoaderA, bool OUT16 = false>
__global__ void __launch_bounds__(256)
gemm_f16cvt_kernel(LoaderA la, const _Float16* __restrict__ B, long long ldb, void* __restrict__ C_, long long ldc,
                   int M, int N, int K, const float* __restrict__ bias, const float* __restrict__ R, long long ldr,
                   int act, int tiles_n, long long a_bs, long long b_bs, long long c_bs) {
  constexpr bool AH = std::is_same<LoaderA, DenseH>::value;
  constexpr bool AD = std::is_same<LoaderA, DualA<true>>::value || std::is_same<LoaderA, DualA<false>>::value;   // two sources along K (gemm_loaders.h)
  constexpr bool AD_F32 = std::is_same<LoaderA, DualA<false>>::value;
  float* C = reinterpret_cast<float*>(C_);                   // (OUT16: only passed on to the epilogue as void*)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_LD = BM * 8 / 256, B_LD = BN * 8 / 256;   // 8-element chunks per thread per K tile
  __shared__ __attribute__((aligned(16))) _Float16 As[BM * LDS_ROW];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * LDS_ROW];

  if (gridDim.y > 1) {   // batched: independent problems along blockIdx.y
    la.advance((long long)blockIdx.y * a_bs);
    B += (long long)blockIdx.y * b_bs;
    C = OUT16 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(C) + (long long)blockIdx.y * c_bs) : C + (long long)blockIdx.y * c_bs;
    if (R) R += (long long)blockIdx.y * c_bs;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bn = (int)(bid % tiles_n) * BN;
  const int bm = (int)(bid / tiles_n) * BM;
  const int srow = tid >> 3, scol = (tid & 7) * 8;

  // staging registers of one K tile; the 64x64 instantiation (92 VGPRs) keeps TWO tiles in flight (set 0 / set 1 alternate), the
  // 128x128 one (208+ VGPRs) one: with 64-column tiles the kernel is bound by the latency of these loads, not by the MFMAs
  struct Stage { float4 pa[A_LD][2]; bool oka[A_LD][2]; uint4 pb[B_LD]; bool okb[B_LD]; bool raw; };   // raw (DualA): pa[.][0] holds 8 fp16 values
  typename LoaderA::RowCtx rca[A_LD];                     // the staged rows of this thread, decomposed once (gemm_loaders.h)
#pragma unroll
  for (int i = 0; i < A_LD; ++i) rca[i] = la.row(bm + srow + i * 32);
  auto gload = [&](Stage& st, int k0) {
    const int k = k0 + scol;
    if constexpr (AD) {
      st.raw = !AD_F32 || k0 < la.K1;                          // wave-uniform (K1 % 64 == 0: the whole K tile lies in one source)
      if (st.raw) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
          const uint4 u = la.load8(rca[i], k, st.oka[i][0]);
          st.pa[i][0] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
      } else if constexpr (AD_F32) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
          st.pa[i][0] = la.load4(rca[i], k, st.oka[i][0]);
          st.pa[i][1] = la.load4(rca[i], k + 4, st.oka[i][1]);
        }
      }
    } else if constexpr (AH) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const uint4 u = la.load8(rca[i], k, st.oka[i][0]);
        st.pa[i][0] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));   // raw bits
      }
    } else {
    const auto kc0 = la.kctx(k), kc1 = la.kctx(k + 4);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      st.pa[i][0] = la.load(rca[i], kc0, st.oka[i][0]);
      st.pa[i][1] = la.load(rca[i], kc1, st.oka[i][1]);
    }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int n = bn + srow + i * 32;
      st.okb[i] = n < N && k < K;
      st.pb[i] = *reinterpret_cast<const uint4*>(B + (st.okb[i] ? (long long)n * ldb + k : 0));
    }
  };
  auto lstore = [&](const Stage& st) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if (AH || (AD && st.raw)) {
        const bool ok = st.oka[i][0];
        const float4 v = st.pa[i][0];
        *reinterpret_cast<uint4*>(&As[(srow + i * 32) * LDS_ROW + scol]) =
            make_uint4(ok ? __float_as_uint(v.x) : 0u, ok ? __float_as_uint(v.y) : 0u, ok ? __float_as_uint(v.z) : 0u, ok ? __float_as_uint(v.w) : 0u);
      } else if constexpr (!AH)
      *reinterpret_cast<uint4*>(&As[(srow + i * 32) * LDS_ROW + scol]) = cvt8(st.oka[i][0], st.pa[i][0], st.oka[i][1], st.pa[i][1]);
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<uint4*>(&Bs[(srow + i * 32) * LDS_ROW + scol]) =
          make_uint4(st.okb[i] ? st.pb[i].x : 0u, st.okb[i] ? st.pb[i].y : 0u, st.okb[i] ? st.pb[i].z : 0u, st.okb[i] ? st.pb[i].w : 0u);
  };

  const int r32 = lane & 31, h = lane >> 5;
  const bool vec_ok = ovis::epilogue_vec_ok(C, ldc, bias, R, ldr);
  const bool pre = !OUT16 && vec_ok && bn + BN <= N && (bias || R);       // accumulators start at bias + residual (gemm_epilogue.h)
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (pre) {
        ovis::acc_init_tile(acc[i][j], min((long long)bm + wr * (BM / 2) + i * 32 + r32, (long long)M - 1),
                            bn + wc * (BN / 2) + j * 32, h, bias, R, ldr);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    }

  const int nk = (K + BKH - 1) / BKH;
  auto compute = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f16x8*>(&As[(wr * (BM / 2) + i * 32 + r32) * LDS_ROW + h * 32 + s * 8]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(&Bs[(wc * (BN / 2) + j * 32 + r32) * LDS_ROW + h * 32 + s * 8]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);   // roles swapped
    }
  };
  Stage s0;
  gload(s0, 0);
  if constexpr (BM == 64) {
    Stage s1;
    gload(s1, BKH);                                        // (past K: clamped addresses, never stored)
    for (int kt = 0; kt < nk; kt += 2) {
      __syncthreads();
      lstore(s0);
      __syncthreads();
      gload(s0, (kt + 2) * BKH);
      compute();
      if (kt + 1 < nk) {
        __syncthreads();
        lstore(s1);
        __syncthreads();
        gload(s1, (kt + 3) * BKH);
        compute();
      }
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      lstore(s0);
      __syncthreads();
      if (kt + 1 < nk) gload(s0, (kt + 1) * BKH);
      compute();
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * (BM / 2) + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      ovis::epilogue_tile<OUT16>(acc[i][j], m, m < M, bn + wc * (BN / 2) + j * 32, h, N, C, ldc, pre ? nullptr : bias,
                                 pre ? nullptr : R, ldr, act, vec_ok);
  }
}

int g_cvt_small_n = 2;             // 0 = 128x128 tiles from 256 tiles on (round 2); 1 = 64x64 for N <= 64; 2 (default) = also below 1024 tiles of 128
                                   // (ResNet-50 on 5 x 720p: 4.70 / 4.42 / 4.41 ms, tools/bench_backbone.py, profiles/r03/backbone_tiles.txt)

template <typename LoaderA, bool OUT16 = false>
int launch(LoaderA la, const _Float16* B, long long ldb, void* C, long long ldc, int M, int N, int K, const float* bias,
           const float* R, long long ldr, int act, hipStream_t stream, int batch = 1, long long a_bs = 0, long long b_bs = 0,
           long long c_bs = 0) {
  const long long blocks128 = (long long)ovis::cdiv(M, 128) * ovis::cdiv(N, 128) * batch;
  // N <= 64 (ResNet res2: 256 -> 64, 3x3 64 -> 64, stem): a 128-column tile would multiply half of its columns for nothing, and the
  // 64x64 instantiation (92 VGPRs, five workgroups per CU instead of two) keeps more of the K loop's loads in flight
  if (blocks128 >= (g_cvt_small_n == 2 ? 1024 : 256) && (N > 64 || !g_cvt_small_n)) {
    const int tm = ovis::cdiv(M, 128), tn = ovis::cdiv(N, 128);
    hipLaunchKernelGGL((gemm_f16cvt_kernel<128, 128, LoaderA, OUT16>), dim3(tm * tn, batch), dim3(256), 0, stream, la, B, ldb, C, ldc,
                       M, N, K, bias, R, ldr, act, tn, a_bs, b_bs, c_bs);
  } else {
    const int tm = ovis::cdiv(M, 64), tn = ovis::cdiv(N, 64);
    hipLaunchKernelGGL((gemm_f16cvt_kernel<64, 64, LoaderA, OUT16>), dim3(tm * tn, batch), dim3(256), 0, stream, la, B, ldb, C, ldc, M,
                       N, K, bias, R, ldr, act, tn, a_bs, b_bs, c_bs);
  }
  return ovis::check_launch("gemm_f16cvt");
}

}  // namespace

extern "C" int ovis_gemm_nt_f32a_f16w(const float* A, long long lda, const void* B16, long long ldb, float* C, long long ldc,
                                      int M, int N, int K, const float* bias, const float* residual, long long ldr,
                                      int act, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B16 && C, "gemm_nt_f32a_f16w: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt_f32a_f16w: non-positive size");
  OVIS_REQUIRE(K % 8 == 0 && lda % 4 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N,
               "gemm_nt_f32a_f16w: need K %% 8 == 0, lda %% 4 == 0, ldb %% 8 == 0");
  OVIS_REQUIRE((((uintptr_t)A | (uintptr_t)B16) & 15) == 0, "gemm_nt_f32a_f16w: A/B must be 16-byte aligned");
  OVIS_REQUIRE(act >= 0 && act <= 3, "gemm_nt_f32a_f16w: unknown activation %d", act);
  OVIS_REQUIRE(!residual || ldr >= N, "gemm_nt_f32a_f16w: residual leading dimension too small");
  return launch(DenseA<true>{A, lda, M, K}, (const _Float16*)B16, ldb, C, ldc, M, N, K, bias, residual, ldr, act,
                (hipStream_t)stream);
}

extern "C" int ovis_gemm_nt_f32a_f16w_batched(const float* A, long long lda, long long a_bs, const void* B16, long long ldb,
                                              long long b_bs, float* C, long long ldc, long long c_bs, int batch, int M,
                                              int N, int K, const float* bias, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B16 && C, "gemm_nt_f32a_f16w_batched: null pointer");
  OVIS_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0, "gemm_nt_f32a_f16w_batched: non-positive size");
  OVIS_REQUIRE(K % 8 == 0 && lda % 4 == 0 && ldb % 8 == 0 && a_bs % 4 == 0 && b_bs % 8 == 0 && lda >= K && ldb >= K &&
                   ldc >= N && (((uintptr_t)A | (uintptr_t)B16) & 15) == 0,
               "gemm_nt_f32a_f16w_batched: alignment (K %% 8, lda %% 4, ldb %% 8, strides, 16-byte pointers)");
  OVIS_REQUIRE(act >= 0 && act <= 3, "gemm_nt_f32a_f16w_batched: unknown activation %d", act);
  return launch(DenseA<true>{A, lda, M, K}, (const _Float16*)B16, ldb, C, ldc, M, N, K, bias, nullptr, 0, act,
                (hipStream_t)stream, batch, a_bs, b_bs, c_bs);
}

extern "C" int ovis_conv2d_nhwc_f32a_f16w(const float* x, const void* w16, float* y, int N, int H, int W, int Cin, int Cout,
                                          int KH, int KW, int stride, int pad, const float* bias, const float* residual,
                                          int act, ovis_stream_t stream) {
  OVIS_REQUIRE(x && w16 && y, "conv2d_nhwc_f32a_f16w: null pointer");
  OVIS_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               "conv2d_nhwc_f32a_f16w: bad geometry");
  OVIS_REQUIRE(Cin % 4 == 0 && (KH * KW * Cin) % 8 == 0, "conv2d_nhwc_f32a_f16w: need Cin %% 4 == 0 and KH*KW*Cin %% 8 == 0");
  OVIS_REQUIRE((((uintptr_t)x | (uintptr_t)w16) & 15) == 0, "conv2d_nhwc_f32a_f16w: x/w must be 16-byte aligned");
  OVIS_REQUIRE(act >= 0 && act <= 3, "conv2d_nhwc_f32a_f16w: unknown activation %d", act);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  OVIS_REQUIRE(OH > 0 && OW > 0, "conv2d_nhwc_f32a_f16w: empty output");
  const long long M = (long long)N * OH * OW;
  OVIS_REQUIRE(M < (1ll << 31), "conv2d_nhwc_f32a_f16w: too many output pixels");
  const int K = KH * KW * Cin;
  ConvA la{x, ConvGeom{H, W, Cin, OH, OW, KH, KW, stride, pad}, (int)M, K};
  return launch(la, (const _Float16*)w16, (long long)K, y, (long long)Cout, (int)M, Cout, K, bias, residual, (long long)Cout,
                act, (hipStream_t)stream);
}

// General form for the fp16-storage path of the backbone: A f32 (rounded to fp16 while staged) or already fp16, C f32 (+ f32 residual) or
// fp16 (no residual).  Same kernel, same arithmetic: fp16 operands, f32 accumulation.
extern "C" int ovis_gemm_nt_x16(const void* A, int a_f16, long long lda, const void* B16, long long ldb, void* C, int c_f16, long long ldc,
                                int M, int N, int K, const float* bias, const float* residual, long long ldr, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B16 && C, "gemm_nt_x16: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N && lda % (a_f16 ? 8 : 4) == 0,
               "gemm_nt_x16: need K %% 8 == 0, ldb %% 8 == 0, lda %% 4 (f32) / 8 (fp16) == 0");
  OVIS_REQUIRE((((uintptr_t)A | (uintptr_t)B16 | (uintptr_t)C) & 15) == 0, "gemm_nt_x16: A / B / C must be 16-byte aligned");
  OVIS_REQUIRE(act >= 0 && act <= 3 && !(c_f16 && residual) && (!residual || ldr >= N), "gemm_nt_x16: bad activation / an fp16 output takes no residual");
  OVIS_REQUIRE(!c_f16 || ldc % 8 == 0, "gemm_nt_x16: fp16 output rows must be 16-byte aligned (ldc %% 8 == 0)");
  hipStream_t s = (hipStream_t)stream;
  const _Float16* B = (const _Float16*)B16;
  if (a_f16) {
    DenseH la{(const _Float16*)A, lda, M, K};
    return c_f16 ? launch<DenseH, true>(la, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, act, s)
                 : launch<DenseH, false>(la, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, s);
  }
  DenseA<true> la{(const float*)A, lda, M, K};
  return c_f16 ? launch<DenseA<true>, true>(la, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, act, s)
               : launch<DenseA<true>, false>(la, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, s);
}

// conv3 + projection shortcut of a bottleneck as one GEMM over the concatenated K axis (DualA, gemm_loaders.h): y = act([A1 | A2] B^T + bias),
// B16 [N, K1 + K2] = [w3 | w_shortcut], bias = b3 + b_shortcut.  _2a: both sources dense fp16 (res2.0: conv2's output and the pooled stem
// output); _pair: second source = the f32 block input x [T, H, W, C2] read at the pixels (s oy, s ox) (the stride-s 1x1 shortcut of res3-5.0).
extern "C" int ovis_gemm_nt_x16_2a(const void* A1_f16, long long lda1, int K1, const void* A2_f16, long long lda2, int K2, const void* B16,
                                   long long ldb, float* C, long long ldc, int M, int N, const float* bias, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(A1_f16 && A2_f16 && B16 && C, "gemm_nt_x16_2a: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K1 > 0 && K2 > 0 && K1 % 64 == 0 && K2 % 8 == 0 && lda1 >= K1 && lda2 >= K2 && lda1 % 8 == 0 && lda2 % 8 == 0 &&
               ldb >= K1 + K2 && ldb % 8 == 0 && ldc >= N, "gemm_nt_x16_2a: need K1 %% 64 == 0, K2 %% 8 == 0, leading dimensions multiples of 8");
  OVIS_REQUIRE((((uintptr_t)A1_f16 | (uintptr_t)A2_f16 | (uintptr_t)B16 | (uintptr_t)C) & 15) == 0 && act >= 0 && act <= 3, "gemm_nt_x16_2a: alignment / activation");
  DualA<true> la{(const _Float16*)A1_f16, lda1, K1, A2_f16, lda2, 0, 0, 0, 0, 0, 0, M, K1 + K2};
  return launch<DualA<true>, false>(la, (const _Float16*)B16, ldb, C, ldc, M, N, K1 + K2, bias, nullptr, 0, act, (hipStream_t)stream);
}

extern "C" int ovis_conv1x1_pair_x16(const void* A1_f16, int K1, const float* x2, int T, int H, int W, int C2, int stride, const void* B16,
                                     float* y, int N, const float* bias, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(A1_f16 && x2 && B16 && y, "conv1x1_pair_x16: null pointer");
  OVIS_REQUIRE(T > 0 && H > 0 && W > 0 && N > 0 && K1 > 0 && K1 % 64 == 0 && C2 > 0 && C2 % 8 == 0 && (stride == 1 || stride == 2),
               "conv1x1_pair_x16: need K1 %% 64 == 0, C2 %% 8 == 0, stride 1 / 2");
  OVIS_REQUIRE((((uintptr_t)A1_f16 | (uintptr_t)x2 | (uintptr_t)B16 | (uintptr_t)y) & 15) == 0 && act >= 0 && act <= 3, "conv1x1_pair_x16: alignment / activation");
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const long long M = (long long)T * OH * OW;
  OVIS_REQUIRE(M < (1ll << 31), "conv1x1_pair_x16: too many output pixels");
  DualA<false> la{(const _Float16*)A1_f16, (long long)K1, K1, x2, 0, H, W, C2, OH, OW, stride, (int)M, K1 + C2};
  return launch<DualA<false>, false>(la, (const _Float16*)B16, (long long)(K1 + C2), y, (long long)N, (int)M, N, K1 + C2, bias, nullptr, 0, act,
                                     (hipStream_t)stream);
}

// ovis_conv2d_nhwc_f32a_f16w with the result written as fp16 (the stem of the fp16-storage backbone; no residual)
extern "C" int ovis_conv2d_nhwc_f32a_f16w_o16(const float* x, const void* w16, void* y_f16, int N, int H, int W, int Cin, int Cout, int KH,
                                              int KW, int stride, int pad, const float* bias, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(x && w16 && y_f16, "conv2d_nhwc_f32a_f16w_o16: null pointer");
  OVIS_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "conv2d_nhwc_f32a_f16w_o16: bad geometry");
  OVIS_REQUIRE(Cin % 4 == 0 && (KH * KW * Cin) % 8 == 0 && Cout % 8 == 0, "conv2d_nhwc_f32a_f16w_o16: need Cin %% 4 == 0, KH*KW*Cin %% 8 == 0, Cout %% 8 == 0");
  OVIS_REQUIRE((((uintptr_t)x | (uintptr_t)w16 | (uintptr_t)y_f16) & 15) == 0, "conv2d_nhwc_f32a_f16w_o16: 16-byte alignment");
  OVIS_REQUIRE(act >= 0 && act <= 3, "conv2d_nhwc_f32a_f16w_o16: unknown activation %d", act);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  OVIS_REQUIRE(OH > 0 && OW > 0, "conv2d_nhwc_f32a_f16w_o16: empty output");
  const long long M = (long long)N * OH * OW;
  OVIS_REQUIRE(M < (1ll << 31), "conv2d_nhwc_f32a_f16w_o16: too many output pixels");
  const int K = KH * KW * Cin;
  ConvA la{x, ConvGeom{H, W, Cin, OH, OW, KH, KW, stride, pad}, (int)M, K};
  return launch<ConvA, true>(la, (const _Float16*)w16, (long long)K, y_f16, (long long)Cout, (int)M, Cout, K, bias, nullptr, 0, act, (hipStream_t)stream);
}

extern "C" int ovis_f16cvt_small_n(int mode) { g_cvt_small_n = mode; return OVIS_OK; }   // 0 old rule, 1 N <= 64 -> 64x64, 2 also < 1024 tiles -> 64x64   // lab / tests only
