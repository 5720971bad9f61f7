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

  const int r32 = lane & 31, h = lane >> 5;
  const bool vec_ok = ovis::epilogue_vec_ok(Cv, ldc, bias, R, ldr);
  const bool pre = vec_ok && bn + BN <= N && (bias || R);       // accumulators start at bias + residual (gemm_epilogue.h)
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

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * (BM / 2) + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      ovis::epilogue_tile<OUT_F16>(acc[i][j], m, m < M, bn + wc * (BN / 2) + j * 32, h, N, Cv, ldc, pre ? nullptr : bias,
                                   pre ? nullptr : R, ldr, act, vec_ok);
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

  const int r32 = lane & 31, h = lane >> 5;
  const bool vec_ok = ovis::epilogue_vec_ok(Cv, ldc, bias, R, ldr);
  const bool pre = vec_ok && bn + BN <= N && (bias || R);       // accumulators start at bias + residual (gemm_epilogue.h)
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (pre) {
        ovis::acc_init_tile(acc[i][j], min((long long)bm + wr * 64 + i * 32 + r32, (long long)M - 1), bn + wc * 64 + j * 32, h,
                            bias, R, ldr);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    }

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

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * 64 + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      ovis::epilogue_tile<OUT_F16>(acc[i][j], m, m < M, bn + wc * 64 + j * 32, h, N, Cv, ldc, pre ? nullptr : bias,
                                   pre ? nullptr : R, ldr, act, vec_ok);
  }
}

// ---- 256x256 tile, 8 wavefronts, 2-stage LDS-DMA double buffer of 64-deep K steps (K % 64 == 0) -------------------
// The 128x128 kernels above move 1 byte per 64 flop through L2 (>9 TB/s at 600 TF): tile-size bound.  This kernel
// doubles the arithmetic intensity (128 flop/B).  A stage holds a 64-deep K step so every DMA'd row segment is one
// FULL 128-byte cache line (a first version with 32-deep stages issued two half-line L2 requests per line and measured
// 1.7x the necessary L2 traffic).  512 threads = 8 waves as 2(M) x 4(N), each 128x64 = 4x2 MFMA 32x32 tiles
// (128 accumulator registers), one workgroup per CU, 2 x 64 KB stages = 128 KB LDS.  Per stage and SIMD there are
// 2 x 32 MFMAs (2048 cycles) between two barriers; the next stage's DMA is issued at the top of the stage and its
// fragments are software-pipelined one MFMA step ahead, with the DMA issue interleaved among the MFMAs.
// Stage image: 256 A rows then 256 B rows of 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the DMA
// source address and on the fragment read (conflict-free ds_read_b128, SQ_LDS_BANK_CONFLICT == 0).
template <bool OUT_F16>
__global__ void __launch_bounds__(512)
gemm_f16_256_kernel(const _Float16* __restrict__ A, long long lda, const _Float16* __restrict__ B, long long ldb,
                    void* __restrict__ Cv, long long ldc, int M, int N, int K, const float* __restrict__ bias,
                    const float* __restrict__ R, long long ldr, int act, int tiles_n, int n_tiles) {
  constexpr int BM = 256, BN = 256, BKS = 64;                  // halfs per stage along K
  constexpr int STAGE = (BM + BN) * BKS;                       // halfs per stage (64 KB)
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                     // 2 x 4 waves
  // PERSISTENT: gridDim.x workgroups (one per CU) walk the tile list; the 32 workgroups of an XCD (blockIdx % 8) take
  // 32 consecutive tiles per round (N-tile fastest), so they share their A tiles / the B panel through that XCD's L2.
  const int nblk = gridDim.x;
  int tile = (int)ovis::xcd_remap(blockIdx.x, nblk);

  const int lrow = lane >> 3, pch = lane & 7;
  const _Float16* src[8];
  auto set_src = [&](int t) {
    const int bn_ = (t % tiles_n) * BN, bm_ = (t / tiles_n) * BM;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int blk = wave * 8 + i;                            // 8-row block of the stage (0-31: A, 32-63: B)
      const int row = (blk & 31) * 8 + lrow;
      const int lch = pch ^ ((row >> 1) & 7);
      if (blk < 32) src[i] = A + (long long)min(bm_ + row, M - 1) * lda + lch * 8;
      else src[i] = B + (long long)min(bn_ + row, N - 1) * ldb + lch * 8;
    }
  };
  auto dma = [&](int slot, int k0) {
    _Float16* base = lds + slot * STAGE + wave * 8 * 8 * BKS;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + i * 8 * BKS), 16, 0, 0);
  };

  const int r32 = lane & 31, h = lane >> 5;
  int aoff[4], boff[2];
  const int sw = (r32 >> 1) & 7;                               // tile row bases are multiples of 32
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (wr * 128 + i * 32 + r32) * BKS;
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = BM * BKS + (wc * 64 + j * 32 + r32) * BKS;
  int coff[4];                                                 // step s reads logical chunk 2s + h
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) coff[s4] = ((2 * s4 + h) ^ sw) << 3;

  f32x16 acc[4][2];
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

  const bool vec_ok = ovis::epilogue_vec_ok(Cv, ldc, bias, R, ldr);
  const int nk = K / BKS;
  f16x8 a0[4], b0[2], a1[4], b1[2];
  int g = 0;                                                   // running stage counter: LDS slot = g & 1
  if (tile < n_tiles) { set_src(tile); dma(0, 0); }
  while (tile < n_tiles) {
    const int bn = (tile % tiles_n) * BN, bm = (tile / tiles_n) * BM;
    // Accumulators start at bias + residual instead of 0 (tiles that lie inside N with 16-byte rows): all 32 residual
    // loads of a lane are in flight together and complete under the wait for the first K stage, and the epilogue is left
    // with conversions and stores only.  (Loaded per 32x32 tile inside the epilogue they cost one exposed memory round
    // trip per tile: +103 us on the N = K = 768 out-projection, +22 us for the bias alone.)
    const bool pre = vec_ok && bn + BN <= N && (bias || R);
    if (pre) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n0 = bn + wc * 64 + j * 32 + 4 * h;
        float4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bv[g] = bias ? *reinterpret_cast<const float4*>(bias + n0 + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long long m = min((long long)bm + wr * 128 + i * 32 + r32, (long long)M - 1);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 rv = R ? *reinterpret_cast<const float4*>(R + m * ldr + n0 + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[i][j][4 * g + 0] = bv[g].x + rv.x; acc[i][j][4 * g + 1] = bv[g].y + rv.y;
            acc[i][j][4 * g + 2] = bv[g].z + rv.z; acc[i][j][4 * g + 3] = bv[g].w + rv.w;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this tile's first stage (issued during the previous epilogue)
    __builtin_amdgcn_s_barrier();
    ldfrag(lds + (g & 1) * STAGE, coff[0], a0, b0);
    for (int kt = 0; kt + 1 < nk; ++kt, ++g) {                 // all stages but the last: prefetch the next K step
      const _Float16* st = lds + (g & 1) * STAGE;
      dma((g + 1) & 1, (kt + 1) * BKS);
      ldfrag(st, coff[1], a1, b1);
      mma(a0, b0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {                            // 1 MFMA : 1 DMA piece
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      ldfrag(st, coff[2], a0, b0);
      mma(a1, b1);
      ldfrag(st, coff[3], a1, b1);
      mma(a0, b0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // own pieces of the next stage landed
      __builtin_amdgcn_s_barrier();                           // ... everyone's; and this stage is fully read
      ldfrag(lds + ((g + 1) & 1) * STAGE, coff[0], a0, b0);
      mma(a1, b1);
    }
    {                                                          // last stage: prefetch the NEXT TILE's first stage instead
      const _Float16* st = lds + (g & 1) * STAGE;
      const int next = tile + nblk;
      if (next < n_tiles) { set_src(next); dma((g + 1) & 1, 0); }
      ldfrag(st, coff[1], a1, b1);
      mma(a0, b0);
      ldfrag(st, coff[2], a0, b0);
      mma(a1, b1);
      ldfrag(st, coff[3], a1, b1);
      mma(a0, b0);
      mma(a1, b1);
      ++g;
    }
    // epilogue: stores overlap the next tile's first-stage DMA (slot g&1); the slot just read is not rewritten before
    // the barrier at the top of the next tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long m = bm + wr * 128 + i * 32 + r32;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        ovis::epilogue_tile<OUT_F16>(acc[i][j], m, m < M, bn + wc * 64 + j * 32, h, N, Cv, ldc, pre ? nullptr : bias,
                                     pre ? nullptr : R, ldr, act, vec_ok);
    }
    tile += nblk;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// y[r, :] (f32, dense) = x[r * ldx + :] (fp16): the class-token rows of an fp16 token tensor
__global__ void __launch_bounds__(256)
cast_f16_f32_rows_kernel(const _Float16* __restrict__ x, long long ldx, float* __restrict__ y, long long rows, int C4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C4) return;
  const long long r = i / C4;
  const int c = (int)(i - r * C4);
  union { uint2 u; _Float16 h[4]; } v;
  v.u = *reinterpret_cast<const uint2*>(x + r * ldx + 4 * c);
  reinterpret_cast<float4*>(y)[i] = make_float4((float)v.h[0], (float)v.h[1], (float)v.h[2], (float)v.h[3]);
}

}  // namespace

namespace ovis {   // gemm_f16_pp.hip: the ping-pong 256x256 kernel for the big CLIP GEMMs
bool gemm_f16_pp_eligible(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K, const float* bias,
                          const float* residual, long long ldr, int out_f16, bool act_is_none);
int gemm_f16_pp_launch(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                       const float* bias, const float* residual, long long ldr, int act, int out_f16, hipStream_t s);
}

extern "C" int ovis_gemm_nt_f16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M,
                                int N, int K, const float* bias, const float* residual, long long ldr, int act,
                                int out_f16, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B && C, "gemm_nt_f16: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt_f16: non-positive size");
  OVIS_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N,
               "gemm_nt_f16: K, lda, ldb must be multiples of 8 halfs");
  OVIS_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm_nt_f16: A/B must be 16-byte aligned");
  OVIS_REQUIRE(act >= 0 && act <= 3, "gemm_nt_f16: unknown activation %d", act);
  OVIS_REQUIRE(!residual || ldr >= N, "gemm_nt_f16: residual leading dimension too small");
  const _Float16* a = reinterpret_cast<const _Float16*>(A);
  const _Float16* b = reinterpret_cast<const _Float16*>(B);
  hipStream_t s = (hipStream_t)stream;
  if (ovis::gemm_f16_pp_eligible(C, lda, ldb, ldc, M, N, K, bias, residual, ldr, out_f16, act == 0))
    return ovis::gemm_f16_pp_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, act, out_f16, s);
  const long long blocks128 = (long long)ovis::cdiv(M, 128) * ovis::cdiv(N, 128);
  const long long blocks256 = (long long)ovis::cdiv(M, 256) * ovis::cdiv(N, 256);
  if (blocks256 >= 256 && K % 64 == 0) {
    const int tm = ovis::cdiv(M, 256), tn = ovis::cdiv(N, 256);
    const int grid = 256;                                       // one persistent workgroup per CU (MI355X: 256 CUs)
    if (out_f16) hipLaunchKernelGGL((gemm_f16_256_kernel<true>), dim3(grid), dim3(512), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn, tm * tn);
    else hipLaunchKernelGGL((gemm_f16_256_kernel<false>), dim3(grid), dim3(512), 0, s, a, lda, b, ldb, C, ldc, M, N, K, bias, residual, ldr, act, tn, tm * tn);
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

extern "C" const char* ovis_gemm_nt_f16_kernel(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K,
                                               const float* bias, const float* residual, long long ldr, int act, int out_f16) {
  // same decision tree as ovis_gemm_nt_f16 (names as rocprofv3 prints them, without the anonymous namespace)
  if (ovis::gemm_f16_pp_eligible(C, lda, ldb, ldc, M, N, K, bias, residual, ldr, out_f16, act == 0))
    // <OUT, ACT, HAS_R, X3, FA, R16> as rocprofv3 prints the instantiation (bench.py joins the PMC summaries on this name)
    return out_f16 ? (act == 0 ? "gemm_f16_pp_kernel<1,0,false,false,false,false>" : act == 1 ? "gemm_f16_pp_kernel<1,1,false,false,false,false>" :
                      act == 2 ? "gemm_f16_pp_kernel<1,2,false,false,false,false>" : "gemm_f16_pp_kernel<1,3,false,false,false,false>")
                   : (residual ? "gemm_f16_pp_kernel<0,0,true,false,false,false>" : "gemm_f16_pp_kernel<0,0,false,false,false,false>");
  const long long blocks128 = (long long)ovis::cdiv(M, 128) * ovis::cdiv(N, 128);
  const long long blocks256 = (long long)ovis::cdiv(M, 256) * ovis::cdiv(N, 256);
  if (blocks256 >= 256 && K % 64 == 0) return out_f16 ? "gemm_f16_256_kernel<true>" : "gemm_f16_256_kernel<false>";
  if (blocks128 >= 128 && K % BKH == 0) return out_f16 ? "gemm_f16_glds_kernel<true>" : "gemm_f16_glds_kernel<false>";
  if (blocks128 >= 128) return out_f16 ? "gemm_f16_kernel<128,128,true>" : "gemm_f16_kernel<128,128,false>";
  return out_f16 ? "gemm_f16_kernel<64,64,true>" : "gemm_f16_kernel<64,64,false>";
}

extern "C" int ovis_cast_f16_to_f32_rows(const void* x_f16, long long ldx, float* y, long long rows, int C, ovis_stream_t stream) {
  OVIS_REQUIRE(x_f16 && y && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldx >= C, "cast_f16_to_f32_rows: C and ldx must be multiples of 4");
  OVIS_REQUIRE((reinterpret_cast<uintptr_t>(x_f16) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "cast_f16_to_f32_rows: alignment");
  hipLaunchKernelGGL(cast_f16_f32_rows_kernel, dim3(ovis::cdiv(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(x_f16), ldx, y, rows, C / 4);
  return ovis::check_launch("cast_f16_to_f32_rows");
}

extern "C" int ovis_cast_f32_to_f16(const float* x, void* y, long long n, ovis_stream_t stream) {
  OVIS_REQUIRE(x && y && n > 0 && n % 4 == 0, "cast_f32_to_f16: n must be a positive multiple of 4");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(ovis::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<uint2*>(y), n / 4);
  return ovis::check_launch("cast_f32_to_f16");
}
