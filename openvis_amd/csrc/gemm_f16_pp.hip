// fp16-input / f32-accumulate GEMM, "ping-pong" schedule for the big CLIP ViT GEMMs (M = crops x tokens >= 65 536 rows).
//
//   C[m,n] = act( sum_k A[m,k]*B[n,k] + bias[n] + R[m,n] ),  A [M,K] / B [N,K] fp16 (K-contiguous), C f32 or fp16.
//   Replaces the cuBLAS calls behind mask_adapted_clip/model.py:238-268 (in_proj / out_proj / c_fc / c_proj of every
//   ResidualAttentionBlock), which the reference runs in fp16 on the GPU (clip.load on cuda; adapter.py:108-111).
//
// Why a second 256x256 kernel.  gemm_f16_256_kernel (gemm_f16.hip) lets all 8 wavefronts of a workgroup run the same
// instruction stream in step: both wavefronts of a SIMD want the matrix pipe at the same time and both read LDS at the
// same time; it measured 0.33 of the dense fp16 peak.  Here the two wavefronts of every SIMD run HALF A PHASE APART:
//
//   * persistent workgroups (one per CU), tile 256x256x64, 8 wavefronts as 2 (M) x 4 (N), 128x64 outputs per wavefront
//     = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (128 accumulator registers);
//   * a K step is cut into 4 phases, one 64x32 output quadrant each: 16 MFMAs fed by 12 / 4 / 8 / 0 ds_read_b128;
//   * every phase is [LDS reads + LDS-DMA issue] s_barrier [16 MFMAs at s_setprio 1] s_barrier, and wavefronts 4-7
//     execute one extra barrier before the loop: while one wavefront of a SIMD issues its 16 MFMAs (256 cycles) the
//     other one reads the next quadrant's fragments and issues the DMA, so the matrix pipe only idles for the barriers;
//   * LDS = 2 K-step buffers x 4 half-tiles (A rows of the two M-halves, B rows of the two N-halves; 16 KB each) filled
//     by global_load_lds_dwordx4 (one half-tile per phase, 2 wave-instructions per wavefront).  The DMA runs 4-5 phases
//     ahead of its first read: the only waits are counted `s_waitcnt vmcnt(8)` (4 half-tiles stay in flight), never 0;
//   * the stream of K steps runs across output tiles: the first 1.5 K steps of the next tile are in flight before the
//     epilogue of the current one starts, and the waits of the first K step after an epilogue leave its stores in flight;
//   * LDS rows are 128 B (a full cache line per DMA'd row segment); the 16-byte chunk index is XOR-swizzled with
//     (row>>1)&7 on the DMA's per-lane SOURCE address and on the fragment read (conflict-free ds_read_b128);
//   * the weight rows of a wavefront's N range are DMA'd in a permuted order (n = 32j + 8q + 4e + r for MFMA row 4q + r of
//     tile 2j + e), so that a lane ends up with 8 consecutive output columns: one 16-byte store per fp16 row segment;
//   * bias lives in LDS for the whole launch (read with ds_read, so the epilogue never touches the vmcnt queue's loads).
//
// Race argument (the hardware orders an LDS-DMA write against a ds_read only through the issuing wavefront's vmcnt wait
// followed by a barrier the reader has passed).  Slots are the intervals between barriers; group G0 = wavefronts 0-3 reads
// in even slots and multiplies in odd ones, G1 = wavefronts 4-7 the other way round.
//   RAW: half-tile X(s) (K step s) is issued >= 4 phases before the phase that reads it; each wavefront executes
//        `vmcnt(8)` at the end of the read segment of the phase BEFORE the reading phase, when exactly 8 younger loads have
//        been issued; the barrier closing that segment precedes every reader's read segment.
//   WAR: a half-tile slot is re-filled >= 2 phases (4 slots) after the phase whose read segment read it; both groups have
//        executed the lgkmcnt wait in front of that phase's MFMAs >= 1 slot before the first DMA instruction is issued.
#include "common.h"
#include "gemm_epilogue.h"
#include <hip/hip_fp16.h>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int PP_HT = 128 * 128;           // bytes of a half-tile: 128 rows x 64 halfs
constexpr int PP_BUF = 4 * PP_HT;          // one K step: A0 A1 B0 B1
constexpr int PP_BIAS = 2 * PP_BUF;        // bias (f32) behind the two buffers
constexpr int PP_LDS = 160 * 1024;
constexpr int PP_MAX_BIAS_N = (PP_LDS - PP_BIAS) / 4;

struct PPArgs {
  const _Float16* A; const _Float16* B; void* C; const float* bias; const float* R;
  long long lda, ldb, ldc, ldr;
  int M, N, K, act;
  int tiles_m, tiles_n, n_tiles;
  int grp_w, grp_rem;                       // raster: column groups of grp_w (+1 for the first grp_rem groups) N tiles
  int desync_ns;                            // start offset spread over the workgroups that own one tile fewer (ns)
  int dbg;                                  // lab only: 1 = skip the epilogue stores, 2 = skip the epilogue arithmetic too
  unsigned long long* stamps;               // lab only: s_memrealtime stamps [workgroup][tile iteration < 16][2 groups][4]
};

#define PP_GLDS(src, dst) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)

template <bool OUT_F16, int ACT, bool HAS_R>
__global__ void __launch_bounds__(512)
gemm_f16_pp_kernel(const PPArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS];   // ONE LDS object (a second one de-pipelines the DMA)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, q = lane >> 4, sw = l15 >> 1;
  const int nblk = gridDim.x;
  const int nk = p.K >> 6;

  for (int i = tid * 4; i < p.N; i += 512 * 4)                   // zeros when there is no bias: no branch in the epilogue
    *reinterpret_cast<float4*>(lds + PP_BIAS + i * 4) =
        p.bias ? *reinterpret_cast<const float4*>(p.bias + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  // logical tile index -> (m tile, n tile): column groups of <= grp_w + 1 N tiles, M-panel-major inside a group; the 32
  // workgroups of an XCD take 32 consecutive logical tiles per round, i.e. a (32 / w) x w block of the output: they
  // share (32 / w + w) operand panels through that XCD's L2 (w ~ 5-6 minimises it; 9-12 N tiles in one row do not).
  auto tile_mn = [&](int L, int& tm, int& tn) {
    const int wb = p.grp_w + 1, big = p.grp_rem * wb * p.tiles_m;
    int n0, w, u;
    if (L < big) { const int g = L / (wb * p.tiles_m); u = L - g * wb * p.tiles_m; n0 = g * wb; w = wb; }
    else { const int L2 = L - big; const int g = L2 / (p.grp_w * p.tiles_m); u = L2 - g * p.grp_w * p.tiles_m;
           n0 = p.grp_rem * wb + g * p.grp_w; w = p.grp_w; }
    tm = u / w; tn = n0 + (u - tm * w);
  };

  // ---- DMA source rows of this lane (byte offsets from A / B; two 8-row groups per half-tile and wavefront) ----
  const int dr = lane >> 3;                                       // row inside the 8-row group
  unsigned offA0[2], offA1[2], offB0[2], offB1[2];
  auto rows_a = [&](int tm, int i, unsigned (&off)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int mr = 16 * (wave & 3) + 8 * g + dr;                // half-tile row = 64 wr + mr
      const int c = (lane & 7) ^ ((q + 4 * g) & 7);               // logical chunk held by this lane's slot: ((row>>1)&7)
      const int m = min(tm * 256 + wr * 128 + i * 64 + mr, p.M - 1);
      off[g] = (unsigned)((long long)m * p.lda * 2 + c * 16);
    }
  };
  auto rows_b = [&](int tn, int j, unsigned (&off)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int i16 = 8 * g + dr, e = wave & 1;                   // half-tile row = 32 (wave>>1) + 16 e + i16
      const int c = (lane & 7) ^ ((q + 4 * g) & 7);
      const int n = min(tn * 256 + (wave >> 1) * 64 + 32 * j + 8 * (i16 >> 2) + 4 * e + (i16 & 3), p.N - 1);
      off[g] = (unsigned)((long long)n * p.ldb * 2 + c * 16);
    }
  };
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  unsigned char* dma_dst = lds + wave * 2048;                      // + buffer + half-tile + g * 1024
  auto issue = [&](const char* base, const unsigned (&off)[2], int kt, int dst_off) {
    PP_GLDS(base + off[0] + kt * 128, dma_dst + dst_off);
    PP_GLDS(base + off[1] + kt * 128, dma_dst + dst_off + 1024);
  };

  // ---- fragment read addresses -----------------------------------------------------------------------------
  const unsigned rd0 = (unsigned)(l15 * 128 + ((q ^ sw) << 4));    // k block 0; k block 1 = ^ 64
  const unsigned a_rd = rd0 + wr * 64 * 128;
  const unsigned b_rd = rd0 + wc * 32 * 128;                      // callers add the B0 slot (2 * PP_HT)

  f32x4 acc[8][4];
  f16x8 af[4][2], bf0[2][2], bf1[2][2];
  auto read_a = [&](unsigned buf, int i) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      af[mb][0] = *reinterpret_cast<const f16x8*>(lds + buf + i * PP_HT + mb * 2048 + a_rd);
      af[mb][1] = *reinterpret_cast<const f16x8*>(lds + buf + i * PP_HT + mb * 2048 + (a_rd ^ 64));
    }
  };
  auto read_b = [&](unsigned buf, int j, f16x8 (&bf)[2][2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      bf[e][0] = *reinterpret_cast<const f16x8*>(lds + buf + j * PP_HT + e * 2048 + b_rd);
      bf[e][1] = *reinterpret_cast<const f16x8*>(lds + buf + j * PP_HT + e * 2048 + (b_rd ^ 64));
    }
  };
  auto mma = [&](int i, int j, const f16x8 (&bf)[2][2]) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[i * 4 + mb][j * 2 + e] =
              __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[e][kb], af[mb][kb], acc[i * 4 + mb][j * 2 + e], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the two DMA cursors: c1 = K step s+1 (half-tiles B1, A1), c2 = K step s+2 (A0, B0) -------------------
  const int first = (int)ovis::xcd_remap(blockIdx.x, nblk);
  int c1_tile = first, c1_kt = 0, c2_tile = first, c2_kt = 0;
  {
    int tm, tn;
    tile_mn(first, tm, tn);
    rows_a(tm, 0, offA0); rows_a(tm, 1, offA1); rows_b(tn, 0, offB0); rows_b(tn, 1, offB1);
  }
  auto advance1 = [&]() {
    if (++c1_kt == nk) {
      c1_kt = 0; c1_tile += nblk;
      if (c1_tile < p.n_tiles) { int tm, tn; tile_mn(c1_tile, tm, tn); rows_a(tm, 1, offA1); rows_b(tn, 1, offB1); }
    }
  };
  auto advance2 = [&]() {
    if (++c2_kt == nk) {
      c2_kt = 0; c2_tile += nblk;
      if (c2_tile < p.n_tiles) { int tm, tn; tile_mn(c2_tile, tm, tn); rows_a(tm, 0, offA0); rows_b(tn, 0, offB0); }
    }
  };
  // (past the last tile the cursors keep re-loading the last tile's rows into slots nobody reads: the vmcnt counts stay uniform)

  if (p.desync_ns > 0) {
    // workgroups that own one tile fewer than the others may start late for free: spreads the epilogue store bursts
    const int rem = p.n_tiles % nblk;
    if (rem && first >= rem) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
      const unsigned long long wait = (unsigned long long)p.desync_ns * (unsigned)(first - rem + 1) / (unsigned)(nblk - rem) / 10;
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
  }

  // prologue: K step 0 (buffer 0) completely, A0/B0 of K step 1 (buffer 1)
  issue(Ab, offA0, 0, 0 * PP_HT); issue(Bb, offB0, 0, 2 * PP_HT);
  advance2();
  issue(Bb, offB1, 0, 3 * PP_HT); issue(Ab, offA1, 0, 1 * PP_HT);
  advance1();
  issue(Ab, offA0, c2_kt, PP_BUF + 0 * PP_HT); issue(Bb, offB0, c2_kt, PP_BUF + 2 * PP_HT);
  advance2();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  // A0, B0 of K step 0
  if (wr == 1) PP_BARRIER();                                       // wavefronts 4-7 run one slot behind
  PP_BARRIER();

  constexpr int NS = OUT_F16 ? 16 : (HAS_R ? 55 : 32);             // younger vm ops of one epilogue (+ residual loads) per lane, 8 + NS <= 63
  unsigned s = 0;                                                    // global K step counter (buffer = s & 1)
  for (int tile = first; tile < p.n_tiles; tile += nblk) {
    int tm, tn;
    tile_mn(tile, tm, tn);
    const int bm = tm * 256, bn = tn * 256;
    const int it = (tile - first) / nblk;
    unsigned long long* st = (p.stamps && it < 16 && (wave & 3) == 0 && lane == 0) ? p.stamps + ((blockIdx.x * 16 + it) * 2 + wr) * 4 : nullptr;
    if (st) st[0] = __builtin_amdgcn_s_memrealtime();
    const int n_lane = bn + wc * 64 + 8 * q;                         // + 32 j: first of this lane's 8 columns
    if constexpr (HAS_R) {                                           // accumulators start at bias + residual (gemm_epilogue.h)
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const long long m = min((long long)bm + wr * 128 + mb * 16 + l15, (long long)p.M - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = min(n_lane + 32 * j, p.N - 8);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float4 rv = *reinterpret_cast<const float4*>(p.R + m * p.ldr + n + 4 * e);
            const float4 bv = *reinterpret_cast<const float4*>(lds + PP_BIAS + (n + 4 * e) * 4);
            acc[mb][2 * j + e] = f32x4{rv.x + bv.x, rv.y + bv.y, rv.z + bv.z, rv.w + bv.w};
          }
        }
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kt = 0; kt < nk; ++kt, ++s) {
      const unsigned cur = (s & 1) * PP_BUF, oth = PP_BUF - cur;
      const bool after_epi = kt == 0 && tile != first;               // this tile's first K step: the epilogue's stores are younger than every load these waits need
      // ---- phase 1: quadrant (0,0) ----
      read_b(cur + 2 * PP_HT, 0, bf0);
      read_a(cur, 0);
      issue(Bb, offB1, c1_kt, oth + 3 * PP_HT);
      if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NS) : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      PP_BARRIER();
      mma(0, 0, bf0);
      PP_BARRIER();
      // ---- phase 2: quadrant (0,1) ----
      read_b(cur + 2 * PP_HT, 1, bf1);
      issue(Ab, offA1, c1_kt, oth + 1 * PP_HT);
      advance1();
      if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NS) : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      PP_BARRIER();
      mma(0, 1, bf1);
      PP_BARRIER();
      // ---- phase 3: quadrant (1,1) ----
      read_a(cur, 1);
      issue(Ab, offA0, c2_kt, cur + 0 * PP_HT);
      PP_BARRIER();
      mma(1, 1, bf1);
      PP_BARRIER();
      // ---- phase 4: quadrant (1,0), no LDS reads ----
      issue(Bb, offB0, c2_kt, cur + 2 * PP_HT);
      advance2();
      if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + NS) : "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      PP_BARRIER();
      mma(1, 0, bf0);
      PP_BARRIER();
    }

    if (st) st[1] = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue: lane = one output row per 16-row block, 8 consecutive columns per column pair ----
    if (p.dbg < 2)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n_lane + 32 * j;
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = 0.f;
      if constexpr (!HAS_R) {
        const int nc = min(n, p.N - 8);
        const float4 b0 = *reinterpret_cast<const float4*>(lds + PP_BIAS + nc * 4);
        const float4 b1 = *reinterpret_cast<const float4*>(lds + PP_BIAS + nc * 4 + 16);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      }
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const long long m = (long long)bm + wr * 128 + mb * 16 + l15;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = acc[mb][2 * j + (e >> 2)][e & 3] + bv[e];
          if constexpr (ACT == 1) v[e] = fmaxf(v[e], 0.f);
          else if constexpr (ACT == 2) v[e] = ovis::quick_gelu(v[e]);
          else if constexpr (ACT == 3) v[e] = ovis::gelu_erf(v[e]);
        }
        const bool ok = m < p.M && n < p.N && p.dbg == 0;                          // N % 8 == 0: a lane's 8 columns are inside or outside together
        if constexpr (OUT_F16) {
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
          if (ok) *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.C) + m * p.ldc + n) = o;
        } else {
          if (ok) {
            float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
      }
    }
    if (st) st[2] = __builtin_amdgcn_s_memrealtime();
  }
  if (wr == 0) PP_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int g_f16_gemm_mode = 1;          // 1: ping-pong kernel for eligible problems, 0: gemm_f16_256_kernel (gemm_f16.hip)
int g_pp_grp = 6;                 // raster: at most this many N tiles per column group
int g_pp_desync_ns = 0;
int g_pp_dbg = 0;
unsigned long long* g_pp_stamps = nullptr;

}  // namespace

namespace ovis {

bool gemm_f16_pp_eligible(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K, const float* bias,
                          const float* residual, long long ldr, int out_f16, bool act_is_none) {
  if (g_f16_gemm_mode != 1) return false;
  if (!((out_f16 && !residual) || (!out_f16 && act_is_none))) return false;   // instantiated combinations
  const long long blocks256 = (long long)cdiv(M, 256) * cdiv(N, 256);
  if (blocks256 < 256 || K % 64 != 0 || K < 128 || N % 8 != 0) return false;
  if ((long long)M * lda * 2 >= (1ll << 32) || (long long)N * ldb * 2 >= (1ll << 32)) return false;   // 32-bit DMA row offsets
  if (bias && (N > PP_MAX_BIAS_N || (reinterpret_cast<uintptr_t>(bias) & 15))) return false;
  if (reinterpret_cast<uintptr_t>(C) & 15) return false;
  if (out_f16 ? (ldc % 8 != 0) : (ldc % 4 != 0)) return false;
  if (residual && ((ldr % 4 != 0) || (reinterpret_cast<uintptr_t>(residual) & 15))) return false;
  return true;   // (act, residual, out dtype) combinations without an instantiation are rejected by gemm_f16_pp_launch's caller check below
}

int gemm_f16_pp_launch(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                       const float* bias, const float* residual, long long ldr, int act, int out_f16, hipStream_t s) {
  PPArgs p;
  p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(B); p.C = C; p.bias = bias; p.R = residual;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = act;
  p.tiles_m = (int)cdiv(M, 256); p.tiles_n = (int)cdiv(N, 256); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  p.desync_ns = g_pp_desync_ns; p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;              // one persistent workgroup per CU (MI355X: 256 CUs)
#define PP_LAUNCH(O, A_, R_) hipLaunchKernelGGL((gemm_f16_pp_kernel<O, A_, R_>), dim3(grid), dim3(512), 0, s, p)
  if (out_f16 && !residual) {
    if (act == 0) PP_LAUNCH(true, 0, false); else if (act == 1) PP_LAUNCH(true, 1, false);
    else if (act == 2) PP_LAUNCH(true, 2, false); else PP_LAUNCH(true, 3, false);
  } else if (!out_f16 && act == 0) {
    if (residual) PP_LAUNCH(false, 0, true); else PP_LAUNCH(false, 0, false);
  } else {
    return fail(OVIS_EINVAL, "gemm_nt_f16 (ping-pong): no instantiation for out_f16=%d act=%d residual=%d", out_f16, act, residual != nullptr);
  }
#undef PP_LAUNCH
  return check_launch("gemm_nt_f16 (ping-pong)");
}

}  // namespace ovis

extern "C" int ovis_set_f16_gemm_mode(int mode, int raster_group, int desync_ns) {
  OVIS_REQUIRE(mode == 0 || mode == 1, "set_f16_gemm_mode: mode must be 0 (gemm_f16_256_kernel) or 1 (ping-pong kernel)");
  OVIS_REQUIRE(raster_group >= 0 && raster_group <= 64 && desync_ns >= 0, "set_f16_gemm_mode: bad tuning value");
  g_f16_gemm_mode = mode;
  if (raster_group > 0) g_pp_grp = raster_group;
  g_pp_desync_ns = desync_ns;
  return OVIS_OK;
}

// lab only (tools/gemm_lab.cpp; not part of include/openvis_hip.h): debug flags and the in-kernel time stamp buffer
extern "C" int ovis_pp_debug(int flags, unsigned long long* stamps) { g_pp_dbg = flags; g_pp_stamps = stamps; return OVIS_OK; }
