// fp16-input / f32-accumulate GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16) — the CLIP ViT GEMMs.
//
//   C[m,n] = act( sum_k A[m,k]*B[n,k] + bias[n] + R[m,n] ),  A,B fp16 (K-contiguous), bias/R f32, C f32 or fp16.
//
// The reference runs CLIP in fp16 on the GPU (clip.load() on cuda; crops are .half()'d, adapter.py:108-111), so the
// CLIP tower's GEMM operands are fp16 here as well; accumulation, bias, residual stream, LayerNorm and softmax stay
// f32 (mask_adapted_clip/model.py:223-229 computes LayerNorm in f32 too).
//
// MI355X mapping: block tile 128x128x64(halfs), 4 wavefronts (2x2) of 64x64 = 2x2 MFMA 32x32 tiles.  The k order
// inside a 64-deep tile is permuted identically for A and B (lane half h owns k in [32h, 32h+32), MFMA step s takes
// its s-th 8-element chunk) so each lane fetches a fragment row segment with 4 ds_read_b128 from rows padded to
// 144 B (conflict-free).  Next tile's dwordx4 global loads are in flight during the 16 MFMAs of the current one.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BKH = 64;             // halfs per K tile
constexpr int LDS_ROW = BKH + 8;    // halfs (144 B)

template <int BM, int BN, bool OUT_F16>
__global__ void __launch_bounds__(256)
gemm_f16_kernel(const _Float16* __restrict__ A, long long lda, const _Float16* __restrict__ B, long long ldb,
                void* __restrict__ Cv, long long ldc, int M, int N, int K, const float* __restrict__ bias,
                const float* __restrict__ R, long long ldr, int act, int tiles_m) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_LD = BM * 8 / 256, B_LD = BN * 8 / 256;   // 16-byte chunks per thread per K tile
  __shared__ __attribute__((aligned(16))) _Float16 As[BM * LDS_ROW];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * LDS_ROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bm = (int)(bid % tiles_m) * BM;
  const int bn = (int)(bid / tiles_m) * BN;
  const int srow = tid >> 3, scol = (tid & 7) * 8;   // halfs

  uint4 pa[A_LD], pb[B_LD];
  bool oka[A_LD], okb[B_LD];     // selects are applied at LDS-store time (keeps the loads in flight under the MFMAs)
  auto gload = [&](int k0) {
    const int k = k0 + scol;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int m = bm + srow + i * 32;
      oka[i] = m < M && k < K;                     // branch-free: clamped address + select (see gemm_f32.hip)
      pa[i] = *reinterpret_cast<const uint4*>(A + (oka[i] ? (long long)m * lda + k : 0));
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int n = bn + srow + i * 32;
      okb[i] = n < N && k < K;
      pb[i] = *reinterpret_cast<const uint4*>(B + (okb[i] ? (long long)n * ldb + k : 0));
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      *reinterpret_cast<uint4*>(&As[(srow + i * 32) * LDS_ROW + scol]) =
          make_uint4(oka[i] ? pa[i].x : 0u, oka[i] ? pa[i].y : 0u, oka[i] ? pa[i].z : 0u, oka[i] ? pa[i].w : 0u);
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<uint4*>(&Bs[(srow + i * 32) * LDS_ROW + scol]) =
          make_uint4(okb[i] ? pb[i].x : 0u, okb[i] ? pb[i].y : 0u, okb[i] ? pb[i].z : 0u, okb[i] ? pb[i].w : 0u);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r32 = lane & 31, h = lane >> 5;
  const int nk = (K + BKH - 1) / BKH;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BKH);

    f16x8 af[TM][4], bf[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const _Float16* p = &As[(wr * (BM / 2) + i * 32 + r32) * LDS_ROW + h * 32];
#pragma unroll
      for (int s = 0; s < 4; ++s) af[i][s] = *reinterpret_cast<const f16x8*>(p + s * 8);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const _Float16* p = &Bs[(wc * (BN / 2) + j * 32 + r32) * LDS_ROW + h * 32];
#pragma unroll
      for (int s = 0; s < 4; ++s) bf[j][s] = *reinterpret_cast<const f16x8*>(p + s * 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = bn + wc * (BN / 2) + j * 32 + r32;
    const bool n_ok = n < N;
    const int nc = n_ok ? n : 0;
    const float bv = bias ? bias[nc] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m0 = bm + wr * (BM / 2) + i * 32 + 4 * h;
      float rv[16];
      if (R) {          // all 16 residual loads issued back to back from clamped addresses (no per-element branch)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (r & 3) + 8 * (r >> 2);
          rv[r] = R[(long long)(m < M ? m : 0) * ldr + nc];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] += acc[i][j][r] + bv;
      if (act == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = fmaxf(rv[r], 0.f);
      } else if (act == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = rv[r] * (1.f / (1.f + expf(-1.702f * rv[r])));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2);
        const float v = rv[r];
        if (n_ok && m < M) {
          if constexpr (OUT_F16) reinterpret_cast<_Float16*>(Cv)[(long long)m * ldc + n] = (_Float16)v;
          else reinterpret_cast<float*>(Cv)[(long long)m * ldc + n] = v;
        }
      }
    }
  }
}

// ---- LDS-DMA variant (K % 64 == 0): global_load_lds_dwordx4 straight into a 2-stage LDS ring -----------------
// The register-staged kernel above is bound by the VGPR->LDS store path (ds_write_b128 ~79 B/clk/CU: 32 KB of tile
// per 512 MFMA cycles).  Here every wave-instruction DMA-copies 64 x 16 B = 8 tile rows (128 B each) into LDS with no
// VGPR round trip.  The LDS image of such a copy is lane-linear, so rows cannot be padded; instead the 16-byte chunk
// index is XOR-swizzled with (row>>1)&7 on the per-lane SOURCE address and on the fragment read (both sides, same
// involution): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots of the 256-B bank row.
// One barrier per K tile: wait own DMA (vmcnt 0) -> barrier -> issue next tile's DMA into the other stage -> MFMAs.
template <bool OUT_F16>
__global__ void __launch_bounds__(256)
gemm_f16_glds_kernel(const _Float16* __restrict__ A, long long lda, const _Float16* __restrict__ B, long long ldb,
                     void* __restrict__ Cv, long long ldc, int M, int N, int K, const float* __restrict__ bias,
                     const float* __restrict__ R, long long ldr, int act, int tiles_m) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int STAGE = (BM + BN) * BKH;                       // halfs per stage (32 KB)
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bm = (int)(bid % tiles_m) * BM;
  const int bn = (int)(bid / tiles_m) * BN;

  // DMA assignment: wave w, instruction i copies tile rows [(4w+i)*8, +8); lane -> (row = L>>3, physical chunk = L&7)
  const int lrow = lane >> 3, pch = lane & 7;
  const _Float16* asrc[4];
  const _Float16* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    const int lch = pch ^ ((row >> 1) & 7);                    // logical chunk stored at this physical slot
    const int ma = min(bm + row, M - 1), nb = min(bn + row, N - 1);
    asrc[i] = A + (long long)ma * lda + lch * 8;
    bsrc[i] = B + (long long)nb * ldb + lch * 8;
  }
  auto dma = [&](int stage, int k0) {
    _Float16* sa = lds + stage * STAGE;
    _Float16* sb = sa + BM * BKH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row0 = (wave * 4 + i) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + k0),
                                       (__attribute__((address_space(3))) void*)(sa + row0 * BKH), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + k0),
                                       (__attribute__((address_space(3))) void*)(sb + row0 * BKH), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r32 = lane & 31, h = lane >> 5;
  // fragment read offsets (halfs) inside a stage: row*64 + ((4h + s) ^ ((row>>1)&7))*8
  int aoff[TM], boff[TN], asw[TM], bsw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) { const int row = wr * 64 + i * 32 + r32; aoff[i] = row * BKH; asw[i] = (row >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < TN; ++j) { const int row = wc * 64 + j * 32 + r32; boff[j] = BM * BKH + row * BKH; bsw[j] = (row >> 1) & 7; }

  const int nk = K / BKH;
  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) dma((kt + 1) & 1, (kt + 1) * BKH);
    const _Float16* st = lds + (kt & 1) * STAGE;
    f16x8 af[TM][4], bf[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) af[i][s] = *reinterpret_cast<const f16x8*>(st + aoff[i] + (((4 * h + s) ^ asw[i]) << 3));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) bf[j][s] = *reinterpret_cast<const f16x8*>(st + boff[j] + (((4 * h + s) ^ bsw[j]) << 3));
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = bn + wc * 64 + j * 32 + r32;
    const bool n_ok = n < N;
    const int nc = n_ok ? n : 0;
    const float bv = bias ? bias[nc] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m0 = bm + wr * 64 + i * 32 + 4 * h;
      float rv[16];
      if (R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (r & 3) + 8 * (r >> 2);
          rv[r] = R[(long long)(m < M ? m : 0) * ldr + nc];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] += acc[i][j][r] + bv;
      if (act == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = fmaxf(rv[r], 0.f);
      } else if (act == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = rv[r] * (1.f / (1.f + expf(-1.702f * rv[r])));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2);
        if (n_ok && m < M) {
          if constexpr (OUT_F16) reinterpret_cast<_Float16*>(Cv)[(long long)m * ldc + n] = (_Float16)rv[r];
          else reinterpret_cast<float*>(Cv)[(long long)m * ldc + n] = rv[r];
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
cast_f32_f16_kernel(const float4* __restrict__ x, uint2* __restrict__ y, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = x[i];
  union { _Float16 h[4]; uint2 u; } o;
  o.h[0] = (_Float16)v.x; o.h[1] = (_Float16)v.y; o.h[2] = (_Float16)v.z; o.h[3] = (_Float16)v.w;
  y[i] = o.u;
}

}  // namespace

extern "C" int ovis_gemm_nt_f16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M,
                                int N, int K, const float* bias, const float* residual, long long ldr, int act,
                                int out_f16, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B && C, "gemm_nt_f16: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt_f16: non-positive size");
  OVIS_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N,
               "gemm_nt_f16: K, lda, ldb must be multiples of 8 halfs");
  OVIS_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm_nt_f16: A/B must be 16-byte aligned");
  OVIS_REQUIRE(act >= 0 && act <= 2, "gemm_nt_f16: unknown activation %d", act);
  OVIS_REQUIRE(!residual || ldr >= N, "gemm_nt_f16: residual leading dimension too small");
  const _Float16* a = reinterpret_cast<const _Float16*>(A);
  const _Float16* b = reinterpret_cast<const _Float16*>(B);
  hipStream_t s = (hipStream_t)stream;
  const long long blocks128 = (long long)ovis::cdiv(M, 128) * ovis::cdiv(N, 128);
  if (blocks128 >= 128 && K % BKH == 0) {
    const int tm = ovis::cdiv(M, 128), tn = ovis::cdiv(N, 128);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_glds_kernel<true>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tm);
    else hipLaunchKernelGGL((gemm_f16_glds_kernel<false>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tm);
  } else if (blocks128 >= 128) {
    const int tm = ovis::cdiv(M, 128), tn = ovis::cdiv(N, 128);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_kernel<128, 128, true>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tm);
    else hipLaunchKernelGGL((gemm_f16_kernel<128, 128, false>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tm);
  } else {
    const int tm = ovis::cdiv(M, 64), tn = ovis::cdiv(N, 64);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_kernel<64, 64, true>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tm);
    else hipLaunchKernelGGL((gemm_f16_kernel<64, 64, false>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tm);
  }
  return ovis::check_launch("gemm_nt_f16");
}

extern "C" int ovis_cast_f32_to_f16(const float* x, void* y, long long n, ovis_stream_t stream) {
  OVIS_REQUIRE(x && y && n > 0 && n % 4 == 0, "cast_f32_to_f16: n must be a positive multiple of 4");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(ovis::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<uint2*>(y), n / 4);
  return ovis::check_launch("cast_f32_to_f16");
}
