// f32 GEMM / implicit-GEMM convolution on the bf16 matrix cores with f32-grade accuracy ("bf16x3").
//
// gfx950's native f32 MFMA (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s; the bf16 MFMA at ~2.5 PFLOP/s.  Every f32
// value is the EXACT sum of three bf16 values (8 + 8 + 8 significand bits = f32's 24, same exponent range):
//     x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (both subtractions are exact)
// so a*b = sum_{p,q} a_p b_q exactly.  The three terms with p + q >= 3 are below 2^-24 |a b| and are dropped; the
// remaining six products (bf16 x bf16 is exact in f32) accumulate in the MFMA's f32 accumulator.  The result carries
// the same rounding class as an f32 fmaf chain (measured: not worse than the native f32 MFMA kernel against an f64
// reference) at 6 bf16 MFMAs per logical product, i.e. a ~417 TFLOP/s ceiling instead of 157.
//
// The split happens while the operands are staged into LDS (v_cvt_pk_bf16_f32 + exact residuals), so HBM traffic and
// the interface (f32 in, f32 out) are unchanged.  Tiling: 128x128x32 per workgroup of 4 waves (64x64 per wave),
// three bf16 planes per operand with 80-byte rows (conflict-free ds_read_b128), register-prefetched next K tile,
// operand roles swapped for the vectorised epilogue (gemm_epilogue.h).
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_loaders.h"

namespace ovis {

using x3_f32x16 = __attribute__((ext_vector_type(16))) float;
using x3_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int X3_BK = 32;
constexpr int X3_ROW = X3_BK + 8;   // bf16 elements per LDS row (80 bytes)

struct X3Planes { uint4 p0, p1, p2; };

__device__ __forceinline__ X3Planes x3_split8(bool ok0, float4 a, bool ok1, float4 b) {
  const float x[8] = {ok0 ? a.x : 0.f, ok0 ? a.y : 0.f, ok0 ? a.z : 0.f, ok0 ? a.w : 0.f,
                      ok1 ? b.x : 0.f, ok1 ? b.y : 0.f, ok1 ? b.z : 0.f, ok1 ? b.w : 0.f};
  union { __bf16 h[8]; uint4 u; } o0, o1, o2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h0 = (__bf16)x[e];
    const float r1 = x[e] - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    o0.h[e] = h0; o1.h[e] = h1; o2.h[e] = (__bf16)r2;
  }
  return X3Planes{o0.u, o1.u, o2.u};
}

// fp16x2 (FH): hi = fp16(x * scale), lo = fp16(x * scale - hi): 11 + 11 significand bits; p2 unused
__device__ __forceinline__ X3Planes x3_split8_f16(bool ok0, float4 a, bool ok1, float4 b, float scale) {
  const float x[8] = {ok0 ? a.x : 0.f, ok0 ? a.y : 0.f, ok0 ? a.z : 0.f, ok0 ? a.w : 0.f,
                      ok1 ? b.x : 0.f, ok1 ? b.y : 0.f, ok1 ? b.z : 0.f, ok1 ? b.w : 0.f};
  union { _Float16 h[8]; uint4 u; } o0, o1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = x[e] * scale;
    const _Float16 h0 = (_Float16)v;
    o0.h[e] = h0; o1.h[e] = (_Float16)(v - (float)h0);
  }
  return X3Planes{o0.u, o1.u, o1.u};
}

// B operand given as three pre-split bf16 planes [3][N][ldb] (constant weights: split once at load, x3_split_kernel)
struct PlanesB {
  const __bf16* P;
  long long ldb, plane;      // row stride and plane stride in elements
  int N, K;
  __device__ __forceinline__ void advance(long long elems) { P += elems; }
};

// NP = 3: the six products above (f32-grade).  NP = 2 ("bf16x2"): planes 0 and 1 only and the three products a1 b0, a0 b1, a0 b0:
// both operands are then carried with 16 significand bits (relative error 2^-17 = 7.6e-6 per operand -- 64 x finer than the fp16
// operands of autocast, 32 x finer than the TF32 cuDNN uses for the reference's f32 convolutions by default) at half the MFMA work.
// FH (NP == 2, pre-split B): "fp16x2" -- the three products on the FP16 matrix cores, operands split into fp16 hi / lo (11 + 11 significand
// bits: every term down to 2^-22 |a b|), activations scaled by fh.a_scale while split, weight planes = fp16 hi / lo of w * fh.w_scale
// (common.h F16x2; gemm_f16_pp.hip explains the scales); the accumulators are scaled back (exactly) before the epilogue.
template <int BM, int BN, typename LoaderA, typename LoaderB = DenseA<true>, int NP = 3, bool FH = false>
__global__ void __launch_bounds__(256)
gemm_f32x3_kernel(LoaderA la, LoaderB lb, float* __restrict__ C, long long ldc, int M, int N, int K,
                  const float* __restrict__ bias, const float* __restrict__ R, long long ldr, int act, int tiles_n,
                  long long a_bs, long long b_bs, long long c_bs, F16x2 fh = F16x2{1.f, 1.f, nullptr}) {
  static_assert(!FH || (NP == 2 && std::is_same<LoaderB, PlanesB>::value), "fp16x2: two pre-split weight planes");
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_IT = BM / 64, B_IT = BN / 64;   // thread stages rows srow + 64 i, 8 consecutive k each
  __shared__ __attribute__((aligned(16))) __bf16 As[NP][BM * X3_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[NP][BN * X3_ROW];

  if (gridDim.y > 1) {
    la.advance((long long)blockIdx.y * a_bs);
    lb.advance((long long)blockIdx.y * b_bs);
    C += (long long)blockIdx.y * c_bs;
    if (R) R += (long long)blockIdx.y * c_bs;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = (int)(bid % tiles_n) * BN;
  const int bm = (int)(bid / tiles_n) * BM;
  // staging row of this thread.  ds_write_b128 is served 8 lanes at a time on 32 banks: lanes 4g .. 4g+3 write the four 16-byte chunks
  // of one 80-byte row, so an octet covers two rows; with CONSECUTIVE rows their bank ranges overlap (20 banks apart: 2-way conflict on
  // every store, SQ_LDS_BANK_CONFLICT = 63 % of the kernel's busy cycles).  Rows 4 apart start 16 banks apart mod 32: conflict-free.
  const int g4 = tid >> 2;
  const int srow = (g4 & ~7) | ((g4 & 1) << 2) | ((g4 >> 1) & 3), scol = (tid & 3) * 8;

  constexpr bool BSPLIT = !std::is_same<LoaderB, PlanesB>::value;   // false: B arrives already split
  float4 pa[A_IT][2], pb[B_IT][2];       // raw prefetch of the tile after next
  bool oka[A_IT][2], okb[B_IT][2];
  X3Planes sa[A_IT], sb[B_IT];           // split planes of the next tile, waiting for the LDS buffer to be free
  X3Planes qb[B_IT];                     // pre-split B: planes in flight
  typename LoaderA::RowCtx rca[A_IT];                     // the staged rows of this thread, decomposed once (gemm_loaders.h)
#pragma unroll
  for (int i = 0; i < A_IT; ++i) rca[i] = la.row(bm + srow + i * 64);
  auto gload = [&](int k0) {
    const int k = k0 + scol;
    const auto kc0 = la.kctx(k), kc1 = la.kctx(k + 4);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      pa[i][0] = la.load(rca[i], kc0, oka[i][0]);
      pa[i][1] = la.load(rca[i], kc1, oka[i][1]);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      if constexpr (BSPLIT) {
        pb[i][0] = lb.load(bn + srow + i * 64, k, okb[i][0]);
        pb[i][1] = lb.load(bn + srow + i * 64, k + 4, okb[i][1]);
      } else {
        const int n = bn + srow + i * 64;
        okb[i][0] = n < lb.N && k < lb.K;                      // K % 8 == 0: a chunk of 8 is all-in or all-out
        const __bf16* p = lb.P + (okb[i][0] ? (long long)n * lb.ldb + k : 0);
        qb[i].p0 = *reinterpret_cast<const uint4*>(p);
        qb[i].p1 = *reinterpret_cast<const uint4*>(p + lb.plane);
        if constexpr (NP == 3) qb[i].p2 = *reinterpret_cast<const uint4*>(p + 2 * lb.plane);
      }
    }
  };
  auto split = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      sa[i] = FH ? x3_split8_f16(oka[i][0], pa[i][0], oka[i][1], pa[i][1], fh.a_scale) : x3_split8(oka[i][0], pa[i][0], oka[i][1], pa[i][1]);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      if constexpr (BSPLIT) sb[i] = x3_split8(okb[i][0], pb[i][0], okb[i][1], pb[i][1]);
      else {
        const bool ok = okb[i][0];
        sb[i].p0 = make_uint4(ok ? qb[i].p0.x : 0u, ok ? qb[i].p0.y : 0u, ok ? qb[i].p0.z : 0u, ok ? qb[i].p0.w : 0u);
        sb[i].p1 = make_uint4(ok ? qb[i].p1.x : 0u, ok ? qb[i].p1.y : 0u, ok ? qb[i].p1.z : 0u, ok ? qb[i].p1.w : 0u);
        if constexpr (NP == 3) sb[i].p2 = make_uint4(ok ? qb[i].p2.x : 0u, ok ? qb[i].p2.y : 0u, ok ? qb[i].p2.z : 0u, ok ? qb[i].p2.w : 0u);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int o = (srow + i * 64) * X3_ROW + scol;
      *reinterpret_cast<uint4*>(&As[0][o]) = sa[i].p0;
      *reinterpret_cast<uint4*>(&As[1][o]) = sa[i].p1;
      if constexpr (NP == 3) *reinterpret_cast<uint4*>(&As[2][o]) = sa[i].p2;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int o = (srow + i * 64) * X3_ROW + scol;
      *reinterpret_cast<uint4*>(&Bs[0][o]) = sb[i].p0;
      *reinterpret_cast<uint4*>(&Bs[1][o]) = sb[i].p1;
      if constexpr (NP == 3) *reinterpret_cast<uint4*>(&Bs[2][o]) = sb[i].p2;
    }
  };

  x3_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r32 = lane & 31, h = lane >> 5;
  const int nk = (K + X3_BK - 1) / X3_BK;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // six products, smallest first
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  auto mma_step = [&](int s) {
    x3_bf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[p][i] = *reinterpret_cast<const x3_bf16x8*>(&As[p][(wr * (BM / 2) + i * 32 + r32) * X3_ROW + h * 16 + s * 8]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[p][j] = *reinterpret_cast<const x3_bf16x8*>(&Bs[p][(wc * (BN / 2) + j * 32 + r32) * X3_ROW + h * 16 + s * 8]);
    }
    // the (i,j) loop is innermost so consecutive MFMAs hit different accumulators
#pragma unroll
    for (int t = (NP == 3 ? 0 : 3); t < 6; ++t)                 // NP == 2: a1 b0, a0 b1, a0 b0
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if constexpr (FH) {
            using f16x8_t = __attribute__((ext_vector_type(8))) _Float16;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, bf[PB[t]][j]), __builtin_bit_cast(f16x8_t, af[PA[t]][i]),
                                                               acc[i][j], 0, 0, 0);
          } else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[PB[t]][j], af[PA[t]][i], acc[i][j], 0, 0, 0);   // roles swapped
  };

  // pipeline: LDS holds tile kt, `sa/sb` hold the split tile kt+1 (produced under the MFMAs of tile kt), `pa/pb` are
  // in flight for tile kt+2 (issued mid-iteration, consumed one full iteration later)
  gload(0);
  split();
  if (nk > 1) gload(X3_BK);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    mma_step(0);
    if (kt + 1 < nk) split();                       // VALU work, independent of the MFMAs around it
    if (kt + 2 < nk) gload((kt + 2) * X3_BK);
    mma_step(1);
  }

  if constexpr (FH) {                              // back to scale 1 (a power of two: exact); a non-finite result = an operand left the fp16 range
    const float inv = 1.f / (fh.a_scale * fh.w_scale);
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {             // (an activation beyond the range makes every output of its row NaN: one element per tile tells)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
        chk = __builtin_fmaf(acc[i][j][0], 0.f, chk);
      }
    if (chk != chk && fh.flag) *fh.flag = 1;
  }
  const bool vec_ok = ((ldc & 3) == 0) && (!R || (ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                      (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * (BM / 2) + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      epilogue_tile<false>(acc[i][j], m, m < M, bn + wc * (BN / 2) + j * 32, h, N, C, ldc, bias, R, ldr, act, vec_ok);
  }
}

int x3_planes();   // gemm_f32.hip: 3 (f32-grade, default) or 2 ("bf16x2", ovis_set_f32_gemm_mode(2))

template <typename LoaderA>
inline void launch_gemm_f32x3(LoaderA la, const float* B, long long ldb, float* C, long long ldc, int M, int N, int K,
                              const float* bias, const float* R, long long ldr, int act, hipStream_t stream, int batch,
                              long long a_bs, long long b_bs, long long c_bs) {
  const int tm = cdiv(M, 128), tn = cdiv(N, 128);
  if (x3_planes() == 2)
    hipLaunchKernelGGL((gemm_f32x3_kernel<128, 128, LoaderA, DenseA<true>, 2>), dim3(tm * tn, batch), dim3(256), 0, stream, la,
                       DenseA<true>{B, ldb, N, K}, C, ldc, M, N, K, bias, R, ldr, act, tn, a_bs, b_bs, c_bs);
  else
  hipLaunchKernelGGL((gemm_f32x3_kernel<128, 128, LoaderA>), dim3(tm * tn, batch), dim3(256), 0, stream, la,
                     DenseA<true>{B, ldb, N, K}, C, ldc, M, N, K, bias, R, ldr, act, tn, a_bs, b_bs, c_bs);
}

// pre-split weights: same kernel, B planes read as 16-byte bf16 chunks (no VALU split for B)
template <typename LoaderA>
inline void launch_gemm_f32x3_w3(LoaderA la, const void* W3, long long ldb, long long plane, float* C, long long ldc, int M, int N,
                                 int K, const float* bias, const float* R, long long ldr, int act, hipStream_t stream) {
  const int tm = cdiv(M, 128), tn = cdiv(N, 128);
  if (x3_planes() == 2)
    hipLaunchKernelGGL((gemm_f32x3_kernel<128, 128, LoaderA, PlanesB, 2>), dim3(tm * tn, 1), dim3(256), 0, stream, la,
                       PlanesB{(const __bf16*)W3, ldb, plane, N, K}, C, ldc, M, N, K, bias, R, ldr, act, tn, 0ll, 0ll, 0ll);
  else
  hipLaunchKernelGGL((gemm_f32x3_kernel<128, 128, LoaderA, PlanesB>), dim3(tm * tn, 1), dim3(256), 0, stream, la,
                     PlanesB{(const __bf16*)W3, ldb, plane, N, K}, C, ldc, M, N, K, bias, R, ldr, act, tn, 0ll, 0ll, 0ll);
}

// fp16x2: H2 = the two fp16 planes of w * fh.w_scale (x2h_split_kernel)
template <typename LoaderA>
inline void launch_gemm_f16x2_h2(LoaderA la, const void* H2, long long ldb, long long plane, float* C, long long ldc, int M, int N,
                                 int K, const float* bias, const float* R, long long ldr, int act, hipStream_t stream, F16x2 fh) {
  const int tm = cdiv(M, 128), tn = cdiv(N, 128);
  if (N <= 64) {   // 64-column tiles (round 5): the 64-channel convolutions of an f32-class ResNet (stem, res2) computed half a 128-column tile of zeros
    hipLaunchKernelGGL((gemm_f32x3_kernel<128, 64, LoaderA, PlanesB, 2, true>), dim3(tm, 1), dim3(256), 0, stream, la,
                       PlanesB{(const __bf16*)H2, ldb, plane, N, K}, C, ldc, M, N, K, bias, R, ldr, act, 1, 0ll, 0ll, 0ll, fh);
    return;
  }
  hipLaunchKernelGGL((gemm_f32x3_kernel<128, 128, LoaderA, PlanesB, 2, true>), dim3(tm * tn, 1), dim3(256), 0, stream, la,
                     PlanesB{(const __bf16*)H2, ldb, plane, N, K}, C, ldc, M, N, K, bias, R, ldr, act, tn, 0ll, 0ll, 0ll, fh);
}

// x [n] f32 -> planes [2][n] fp16: hi = fp16(x * scale), lo = fp16(x * scale - hi)
__global__ void __launch_bounds__(256)
x2h_split_kernel(const float* __restrict__ x, _Float16* __restrict__ planes, long long n, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i] * scale;
  const _Float16 h0 = (_Float16)v;
  planes[i] = h0; planes[n + i] = (_Float16)(v - (float)h0);
}

// x [n] f32 -> planes [3][n] bf16 with x == p0 + p1 + p2 exactly
__global__ void __launch_bounds__(256)
x3_split_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const __bf16 h0 = (__bf16)v;
  const float r1 = v - (float)h0;
  const __bf16 h1 = (__bf16)r1;
  const float r2 = r1 - (float)h1;
  planes[i] = h0; planes[n + i] = h1; planes[2 * n + i] = (__bf16)r2;
}

}  // namespace ovis
