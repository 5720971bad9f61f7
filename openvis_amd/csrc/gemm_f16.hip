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
#include "gemm_epilogue.h"
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
                const float* __restrict__ R, long long ldr, int act, int tiles_n) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_LD = BM * 8 / 256, B_LD = BN * 8 / 256;   // 16-byte chunks per thread per K tile
  __shared__ __attribute__((aligned(16))) _Float16 As[BM * LDS_ROW];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * LDS_ROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bn = (int)(bid % tiles_n) * BN;   // N-tile fastest (A tile shared through the XCD's L2), see gemm_f32.hip
  const int bm = (int)(bid / tiles_n) * BM;
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);   // roles swapped
  }

  const bool vec_ok = ((ldc & 3) == 0) && (!R || (ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cv) & 15) == 0) &&
                      (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * (BM / 2) + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      ovis::epilogue_tile<OUT_F16>(acc[i][j], m, m < M, bn + wc * (BN / 2) + j * 32, h, N, Cv, ldc, bias, R, ldr, act, vec_ok);
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
                     const float* __restrict__ R, long long ldr, int act, int tiles_n) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int STAGE = (BM + BN) * BKH;                       // halfs per stage (32 KB)
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bn = (int)(bid % tiles_n) * BN;   // N-tile fastest (A tile shared through the XCD's L2), see gemm_f32.hip
  const int bm = (int)(bid / tiles_n) * BM;

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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);   // roles swapped
  }

  const bool vec_ok = ((ldc & 3) == 0) && (!R || (ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cv) & 15) == 0) &&
                      (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * 64 + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      ovis::epilogue_tile<OUT_F16>(acc[i][j], m, m < M, bn + wc * 64 + j * 32, h, N, Cv, ldc, bias, R, ldr, act, vec_ok);
  }
}

// ---- 256x256 tile, 8 wavefronts, 4-stage LDS-DMA ring with counted vmcnt (K % 32 == 0) ----------------------------
// The 128x128 kernels above move 1 byte per 64 flop through L2 (>9 TB/s at 600 TF): tile-size bound.  This kernel
// doubles the arithmetic intensity (128 flop/B) and keeps THREE 32-deep K stages of LDS-DMA in flight across raw
// s_barriers (s_waitcnt vmcnt(8) — never 0 in the steady state), so DMA latency (~1-1.5 us under load) hides behind
// 3 x 1024 MFMA cycles per SIMD.  512 threads = 8 waves as 2(M) x 4(N), each 128x64 = 4x2 MFMA 32x32 tiles
// (128 accumulator registers), one workgroup per CU, 4 x 32 KB stages = 128 KB LDS.
// Stage image: 256 A rows then 256 B rows of 64 B; the 16-B chunk index is XOR-swizzled with (row>>2)&3 on the DMA
// source address and on the fragment read (ds_read_b128 groups then cover 16 distinct slots of a 256-B bank row).
template <bool OUT_F16>
__global__ void __launch_bounds__(512)
gemm_f16_256_kernel(const _Float16* __restrict__ A, long long lda, const _Float16* __restrict__ B, long long ldb,
                    void* __restrict__ Cv, long long ldc, int M, int N, int K, const float* __restrict__ bias,
                    const float* __restrict__ R, long long ldr, int act, int tiles_n) {
  constexpr int BM = 256, BN = 256, BKS = 32;                  // halfs per stage along K
  constexpr int STAGE = (BM + BN) * BKS;                       // halfs per stage (32 KB)
  constexpr int NST = 4;
  __shared__ __attribute__((aligned(16))) _Float16 lds[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                     // 2 x 4 waves
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bn = (int)(bid % tiles_n) * BN;   // N-tile fastest (A tile shared through the XCD's L2), see gemm_f32.hip
  const int bm = (int)(bid / tiles_n) * BM;

  // DMA: wave w, instruction i copies the 16-row block (4w + i) of the 32 blocks of a stage (0-15: A, 16-31: B)
  const int lrow = lane >> 2, pch = lane & 3;
  const int lch = pch ^ ((lrow >> 2) & 3);
  const _Float16* src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = wave * 4 + i;
    if (blk < 16) src[i] = A + (long long)min(bm + blk * 16 + lrow, M - 1) * lda + lch * 8;
    else src[i] = B + (long long)min(bn + (blk - 16) * 16 + lrow, N - 1) * ldb + lch * 8;
  }
  auto dma = [&](int stage, int k0) {
    _Float16* base = lds + stage * STAGE + wave * 4 * 16 * BKS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + i * 16 * BKS), 16, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r32 = lane & 31, h = lane >> 5;
  // fragment offsets (halfs) inside a stage; step s reads logical chunk 2s + h at physical (2s+h) ^ ((row>>2)&3)
  int aoff[4], boff[2];
  const int sw = (r32 >> 2) & 3;                               // tile row bases are multiples of 32 -> depends on r32 only
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (wr * 128 + i * 32 + r32) * BKS;
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = BM * BKS + (wc * 64 + j * 32 + r32) * BKS;
  const int c0 = ((0 + h) ^ sw) << 3, c1 = ((2 + h) ^ sw) << 3;

  const int nk = K / BKS;
  dma(0, 0);
  if (nk > 1) dma(1, BKS);
  if (nk > 2) dma(2, 2 * BKS);
  // Software pipeline over the two 16-deep MFMA steps of every stage: the fragments of the NEXT step are fetched
  // from LDS before the MFMAs of the current step are issued, and the stage hand-over (counted vmcnt + raw barrier)
  // sits between step 0 and step 1, so neither LDS latency, LDS-DMA issue nor the barrier drain the matrix pipe.
  auto ldfrag = [&](const _Float16* st, int cc, f16x8 (&af)[4], f16x8 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f16x8*>(st + aoff[i] + cc);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f16x8*>(st + boff[j] + cc);
  };
  auto mma = [&](const f16x8 (&af)[4], const f16x8 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);   // roles swapped
  };
  f16x8 a0[4], b0[2], a1[4], b1[2];
  if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  ldfrag(lds, c0, a0, b0);
  for (int kt = 0; kt < nk; ++kt) {
    const _Float16* st = lds + (kt & 3) * STAGE;
    ldfrag(st, c1, a1, b1);                                   // step-1 fragments of this stage
    if (kt + 3 < nk) dma((kt + 3) & 3, (kt + 3) * BKS);       // ring slot (kt-1)&3: every wave passed its barrier
    mma(a0, b0);
    if (kt + 1 < nk) {
      if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                           // stage kt+1 landed for everyone; stage kt fully read
      ldfrag(lds + ((kt + 1) & 3) * STAGE, c0, a0, b0);       // step-0 fragments of the next stage
    }
    mma(a1, b1);
  }

  const bool vec_ok = ((ldc & 3) == 0) && (!R || (ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cv) & 15) == 0) &&
                      (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long m = bm + wr * 128 + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      ovis::epilogue_tile<OUT_F16>(acc[i][j], m, m < M, bn + wc * 64 + j * 32, h, N, Cv, ldc, bias, R, ldr, act, vec_ok);
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
  const long long blocks256 = (long long)ovis::cdiv(M, 256) * ovis::cdiv(N, 256);
  if (blocks256 >= 256 && K % 32 == 0) {
    const int tm = ovis::cdiv(M, 256), tn = ovis::cdiv(N, 256);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_256_kernel<true>), dim3(tm * tn), dim3(512), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
    else hipLaunchKernelGGL((gemm_f16_256_kernel<false>), dim3(tm * tn), dim3(512), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
  } else if (blocks128 >= 128 && K % BKH == 0) {
    const int tm = ovis::cdiv(M, 128), tn = ovis::cdiv(N, 128);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_glds_kernel<true>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
    else hipLaunchKernelGGL((gemm_f16_glds_kernel<false>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
  } else if (blocks128 >= 128) {
    const int tm = ovis::cdiv(M, 128), tn = ovis::cdiv(N, 128);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_kernel<128, 128, true>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
    else hipLaunchKernelGGL((gemm_f16_kernel<128, 128, false>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
  } else {
    const int tm = ovis::cdiv(M, 64), tn = ovis::cdiv(N, 64);
    if (out_f16) hipLaunchKernelGGL((gemm_f16_kernel<64, 64, true>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
    else hipLaunchKernelGGL((gemm_f16_kernel<64, 64, false>), dim3(tm * tn), dim3(256), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn);
  }
  return ovis::check_launch("gemm_nt_f16");
}

extern "C" int ovis_cast_f32_to_f16(const float* x, void* y, long long n, ovis_stream_t stream) {
  OVIS_REQUIRE(x && y && n > 0 && n % 4 == 0, "cast_f32_to_f16: n must be a positive multiple of 4");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(ovis::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<uint2*>(y), n / 4);
  return ovis::check_launch("cast_f32_to_f16");
}
