// gemm_onewave_lab: the experiment round 4's review asked for (VERDICT.md "Next round" item 4) -- can a K loop with ONE wavefront per
// SIMD (4 wavefronts x 128x128 outputs, 256 accumulator registers of the 512-register budget, LDS fragment reads and the LDS-DMA issue
// software-pipelined between the wave's own MFMAs, one barrier per K step) beat the shipped ping-pong kernel (8 wavefronts, two per SIMD
// half a phase apart)?  The idea behind it: with 512 registers the previous tile's packed result could be stored during the next tile's
// first K steps, hiding the ~5.8 us tile transition.  That only pays if the K loop itself is at least as fast, so this lab measures the K
// loop first: a plain (non-persistent) 256x256x64 kernel, C = A B^T in fp16 / f32 accumulate, fp16 output, simple epilogue.
//
//   build:  make -C openvis_amd/csrc lab1      ->  build/gemm_onewave_lab
//   run  :  build/gemm_onewave_lab              (checks on small-integer operands, then interleaved timing against ovis_gemm_nt_f16)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "../include/openvis_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define OVIS_OKAY(x) do { int r_ = (x); if (r_ != 0) { printf("ovis error %d: %s at %s:%d\n", r_, ovis_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int OW_HT = 256 * 128;          // bytes of one operand's K step: 256 rows x 64 halfs
constexpr int OW_BUF = 2 * OW_HT;         // A | B
constexpr int OW_LDS = 2 * OW_BUF;        // two K steps = 128 KB

#define OW_GLDS(src, dst) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define OW_FENCE() __builtin_amdgcn_sched_barrier(0)

// VARIANT 0: reads + DMA interleaved between the MFMAs (the design under test); 1: no DMA inside the loop (operands of K step 0 re-used:
// what the MFMA + fragment-read stream alone sustains); 2: no MFMAs (DMA + reads + barriers alone)
template <int VARIANT, bool BUF = false>
__global__ void __launch_bounds__(256, 1)
onewave_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, _Float16* __restrict__ C, int M, int N, int K,
               unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OW_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                       // 2 x 2 wavefronts, 128 x 128 outputs each
  const int tiles_n = N / 256;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int nk = K / 64;

  // ---- DMA: this wavefront moves rows 64 w .. 64 w + 63 of A and of B (8 groups of 8 rows each): lane -> (row lane >> 3, slot lane & 7),
  // the slot holds logical chunk slot ^ ((row >> 1) & 7) (swizzle on the SOURCE address, linear LDS destination)
  const int dr = lane >> 3, slot = lane & 7;
  unsigned voA[8], voB[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int row = 64 * wave + 8 * g + dr;
    const int c = slot ^ ((row >> 1) & 7);
    voA[g] = (unsigned)(((long long)(tm * 256 + row) * K) * 2 + c * 16);
    voB[g] = (unsigned)(((long long)(tn * 256 + row) * K) * 2 + c * 16);
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(B);
  unsigned char* dstA = lds + wave * 8192;                        // + buf * OW_BUF + g * 1024
  unsigned char* dstB = lds + OW_HT + wave * 8192;
  // BUF: buffer_load ... lds with the K-step offset in an SGPR (soffset): no per-instruction VALU address arithmetic
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((long long)M * K * 2), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((long long)N * K * 2), 0x00020000);
  auto dma_a = [&](int step, int g) {
    if constexpr (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dstA + (step & 1) * OW_BUF + g * 1024), 16, voA[g], step * 128, 0, 0);
    else OW_GLDS(Ab + voA[g] + step * 128, dstA + (step & 1) * OW_BUF + g * 1024);
  };
  auto dma_b = [&](int step, int g) {
    if constexpr (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dstB + (step & 1) * OW_BUF + g * 1024), 16, voB[g], step * 128, 0, 0);
    else OW_GLDS(Bb + voB[g] + step * 128, dstB + (step & 1) * OW_BUF + g * 1024);
  };

  // ---- fragment reads: mfma_f32_32x32x16_f16 takes, per lane, 8 consecutive k of row (lane & 31): k = 16 ks + 8 (lane >> 5)
  const int r32 = lane & 31, kh = lane >> 5;
  auto frag = [&](int buf, int operand, int row_block, int ks) {   // operand 0 = A (rows 128 wr + 32 i + r32), 1 = B (rows 128 wc + ...)
    const int row = 128 * (operand ? wc : wr) + 32 * row_block + r32;
    const int c = (2 * ks + kh) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const f16x8*>(lds + buf * OW_BUF + operand * OW_HT + row * 128 + c * 16);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: K steps 0 and 1 in flight, step 0 landed
#pragma unroll
  for (int g = 0; g < 8; ++g) { dma_a(0, g); dma_b(0, g); }
  if (nk > 1) {
#pragma unroll
    for (int g = 0; g < 8; ++g) { dma_a(1, g); dma_b(1, g); }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  OW_FENCE();
  unsigned long long t0 = 0;
  if (stamps && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();

  // Rotated loop over the 4 nk k-slices (16 MFMAs each).  Slice (s, ks) multiplies the fragments read during the slice before it and,
  // between its MFMAs, reads the fragments of the NEXT slice (which for ks = 3 lie in the other buffer: K step s + 1).  The step is CLOSED
  // after slice (s, 2): by then every fragment of K step s has been read (lgkmcnt(0)) and this wavefront's DMAs have landed (vmcnt(0)); the
  // barrier behind those two waits therefore (a) frees buffer s & 1 for the DMA of K step s + 2, issued during slices (s, 3) and (s + 1, 0)
  // -- two whole slices before their data is needed -- and (b) publishes K step s + 1.  No fragment read is exposed: slice (s, 3)'s
  // MFMAs run behind the barrier while the first fragments of step s + 1 arrive.
  f16x8 af[2][4], bf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { af[0][i] = frag(0, 0, i, 0); bf[0][i] = frag(0, 1, i, 0); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  OW_FENCE();
  for (int s = 0; s < nk; ++s) {
    const int buf = VARIANT == 1 ? 0 : (s & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      const bool more = ks < 3 || s + 1 < nk;                       // a next slice exists
      const int nbuf = ks < 3 ? buf : (VARIANT == 1 ? 0 : (buf ^ 1)), nks = (ks + 1) & 3;
      // DMA of K step s + 2 -> buffer s & 1: in slice (s, 3) [groups 0..3] and slice (s + 1, 0) [groups 4..7]
      const int dstep = ks == 3 ? s + 2 : s + 1;                    // (in slice (s, 0) the step being loaded is (s - 1) + 2)
      const bool dma_on = VARIANT != 1 && (ks == 3 || (ks == 0 && s >= 1)) && dstep < nk;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = t >> 2, j = t & 3;
        if (VARIANT != 2)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        else
          asm volatile("" :: "v"(af[cur][i]), "v"(bf[cur][j]));      // keep the fragment reads alive (rule 17: ablation must not DCE them)
        // fragment reads of the next slice behind MFMAs 0..7 (done long before the slice ends: the wait below exposes no LDS latency),
        // the DMA instructions behind MFMAs 8..15
        if (t < 8) {
          if (more) {
            if (t < 4) af[nxt][t] = frag(nbuf, 0, t, nks);
            else bf[nxt][t - 4] = frag(nbuf, 1, t - 4, nks);
          }
        } else if (dma_on) {
          const int u = t - 8, g = (ks == 3 ? 0 : 4) + (u >> 1);
          if ((u & 1) == 0) dma_a(dstep, g); else dma_b(dstep, g);
        }
        OW_FENCE();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ks == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      OW_FENCE();
    }
  }
  if (stamps && tid == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }

  // epilogue (plain): lane holds column n = 32 j + r32 of its wavefront's 128, rows 32 i + (r & 3) + 8 (r >> 2) + 4 kh
  _Float16* cp = C + (long long)(tm * 256 + 128 * wr) * N + tn * 256 + 128 * wc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh, n = 32 * j + r32;
        cp[(long long)m * N + n] = (_Float16)acc[i][j][r];
      }
}

// =====================================================================================================================================
// v2: the same K loop, PERSISTENT (one workgroup per CU walks its tiles; the stream of K steps runs across tiles: the DMA cursor is two K
// steps ahead of the multiply, also over a tile change), column-group raster + XCD remap as the shipped kernel, and the tile's result
// leaves as packed fp16 HELD IN REGISTERS (128 of the 512): it is converted between the tiles and STORED BETWEEN THE MFMAs of the next
// tile's first K step.  MFMA operands are swapped against v1 (a = weight rows, b = activation rows) so that a lane owns an output ROW
// (m = 32 i + lane & 31) and, after one v_permlane32_swap per register pair with its partner lane (same row, other k half), 8 consecutive
// columns per 16-byte store.
// =====================================================================================================================================
struct OWArgs { const _Float16* A; const _Float16* B; _Float16* C; int M, N, K, tiles_m, tiles_n, n_tiles, grp_w, grp_rem; unsigned long long* stamps; };
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ unsigned ow_xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// OVERLAP false: the stores are issued right behind the conversion (what hiding them is worth)
template <bool OVERLAP>
__global__ void __launch_bounds__(256, 1)
onewave_persist_kernel(const OWArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OW_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = p.K / 64, nblk = gridDim.x;
  const int first = (int)ow_xcd_remap(blockIdx.x, nblk);
  const int n_my = (p.n_tiles - first + nblk - 1) / nblk;
  auto tile_mn = [&](int L, int& tm, int& tn) {
    const int wb = p.grp_w + 1, big = p.grp_rem * wb * p.tiles_m;
    int n0, w, u;
    if (L < big) { const int g = L / (wb * p.tiles_m); u = L - g * wb * p.tiles_m; n0 = g * wb; w = wb; }
    else { const int L2 = L - big; const int g = L2 / (p.grp_w * p.tiles_m); u = L2 - g * p.grp_w * p.tiles_m; n0 = p.grp_rem * wb + g * p.grp_w; w = p.grp_w; }
    tm = u / w; tn = n0 + (u - tm * w);
  };

  const int dr = lane >> 3, slot = lane & 7;
  unsigned voA[8];                                                  // tile-relative lane offsets (bytes), the same for A and B rows
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int row = 64 * wave + 8 * g + dr;
    const int c = slot ^ ((row >> 1) & 7);
    voA[g] = (unsigned)((long long)row * p.K * 2 + c * 16);
    asm volatile("" : "+v"(voA[g]));
  }
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long long)p.M * p.K * 2), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
  const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)((long long)p.M * p.N * 2), 0x00020000);
  unsigned char* dstA = lds + wave * 8192;
  unsigned char* dstB = lds + OW_HT + wave * 8192;
  // DMA cursor: global K step `cgs` = K step `ck` of this workgroup's tile `ci` (byte offsets of that tile's first rows in A / B)
  int ci = 0, ck = 0, cgs = 0;
  unsigned caoff = 0, cboff = 0;
  auto cursor_tile = [&]() {      // (past the last tile the cursor keeps re-loading the last tile's rows into buffers nobody reads: no branch per DMA)
    int tm, tn; tile_mn(first + (ci < n_my ? ci : n_my - 1) * nblk, tm, tn); caoff = (unsigned)((long long)tm * 256 * p.K * 2); cboff = (unsigned)((long long)tn * 256 * p.K * 2);
  };
  cursor_tile();
  auto dma = [&](int g, int which) {                                // one instruction of the cursor's K step
    const int b = (cgs & 1) * OW_BUF;
    if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dstA + b + g * 1024), 16, voA[g], caoff + ck * 128, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dstB + b + g * 1024), 16, voA[g], cboff + ck * 128, 0, 0);
  };
  auto cursor_next = [&]() { ++cgs; if (++ck == nk) { ck = 0; ++ci; cursor_tile(); } };

  const int r32 = lane & 31, kh = lane >> 5;
  // fragment addresses: the swizzle term (row >> 1) & 7 of row 128 w + 32 rb + r32 is (r32 >> 1) & 7 for every row block, and the chunk
  // 2 ks + kh = (2 ks) ^ kh, so a lane needs one LDS offset per (buffer, operand, k slice) -- 16 registers, kept opaque so that the
  // compiler does not expand them into one register per (row block, ...) combination -- plus the immediate 4096 rb
  unsigned fb[2][2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fb[b][op][ks] = (unsigned)(b * OW_BUF + op * OW_HT + (128 * (op ? wc : wr) + r32) * 128 + (((2 * ks) ^ kh ^ ((r32 >> 1) & 7)) << 4));
        asm volatile("" : "+v"(fb[b][op][ks]));
      }
  auto frag = [&](int buf, int operand, int row_block, int ks) {
    return *reinterpret_cast<const f16x8*>(lds + (buf ? fb[1][operand][ks] : fb[0][operand][ks]) + row_block * 4096);
  };

  f32x16 acc[4][4];                                                 // [i: 32-row block][j: 32-column block]
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  u32x4 pk[4][4][2];                                                // the previous tile's result, packed: [i][j][16-column half] = 8 fp16 of this lane's chunk
  unsigned st_tile = 0;                                             // byte offset of the previous tile's first element in C
  const unsigned lane_c = (unsigned)(((long long)(128 * wr + r32) * p.N + 128 * wc + 8 * kh) * 2);
  auto store_blk = [&](int i, int j, int h) {                        // 16 bytes: row 32 i + r32, columns 32 j + 16 h + 8 kh ...
    __builtin_amdgcn_raw_buffer_store_b128(pk[i][j][h], rsC, lane_c, st_tile + (unsigned)((long long)32 * i * p.N * 2 + (32 * j + 16 * h) * 2), 0);
  };

  // prologue: K step 0 and the first half (groups 0..3) of K step 1.  From then on the schedule is uniform: slice 0 of K step gs issues the
  // SECOND half of step gs + 1 (its buffer was freed by the barrier of step gs - 1), slice 3 -- behind the barrier that closes step gs --
  // the FIRST half of step gs + 2 into the buffer step gs has just finished reading.
#pragma unroll
  for (int g = 0; g < 8; ++g) { dma(g, 0); dma(g, 1); }
  cursor_next();
#pragma unroll
  for (int g = 0; g < 4; ++g) { dma(g, 0); dma(g, 1); }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  OW_FENCE();
  f16x8 af[2][4], bf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { af[0][i] = frag(0, 0, i, 0); bf[0][i] = frag(0, 1, i, 0); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  OW_FENCE();

  const auto rsNull = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0, 0x00020000);   // zero records: the range check drops every store through it
  int gs = 0;                                                       // global K step of the multiply
  // one k-slice = 16 MFMAs; what rides between them is fixed at COMPILE time: FRESH = accumulators start at zero, STORE = 16 of the previous
  // tile's stores; every slice 0 / 3 carries 8 DMA instructions, every slice reads the next slice's fragments (past the last K step they
  // come from a buffer nobody filled: never multiplied)
  auto slice = [&](auto ks_t, auto fresh_t, auto store_t, const auto& rs_store) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_t)::value;
    constexpr bool FRESH = decltype(fresh_t)::value, STORE = decltype(store_t)::value;
    constexpr int cur = ks & 1, nxt = cur ^ 1, nks = (ks + 1) & 3;
    const int buf = gs & 1, nbuf = ks < 3 ? buf : (buf ^ 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int i = t >> 2, j = t & 3;
      if constexpr (FRESH) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur][j], af[cur][i], zero, 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
      if (t < 8) {
        if (t < 4) af[nxt][t] = frag(nbuf, 0, t, nks);
        else bf[nxt][t - 4] = frag(nbuf, 1, t - 4, nks);
      } else if constexpr (ks == 0 || ks == 3) {
        const int u = t - 8;
        dma((ks == 3 ? 0 : 4) + (u >> 1), u & 1);
      }
      if constexpr (STORE) {
        const int b = (ks - 1) * 8 + (t >> 1), bi = b >> 2, bj = b & 3, h = t & 1;
        __builtin_amdgcn_raw_buffer_store_b128(pk[bi][bj][h], rs_store, lane_c, st_tile + (unsigned)((long long)32 * bi * p.N * 2 + (32 * bj + 16 * h) * 2), 0);
      }
      OW_FENCE();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OW_FENCE();
  };
  using T = std::true_type; using F = std::false_type;
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
  for (int ti = 0; ti < n_my; ++ti) {
    unsigned long long t_begin = 0;
    if (p.stamps && tid == 0 && ti < 16) t_begin = __builtin_amdgcn_s_memrealtime();
    // ---- first K step of the tile: fresh accumulators; the previous tile's 32 stores ride in slices 1 and 2 (tile 0: dropped by rsNull) ----
    {
      const auto rs_st = (OVERLAP && ti > 0) ? rsC : rsNull;
      slice(K0{}, T{}, F{}, rsNull); cursor_next();
      if constexpr (OVERLAP) { slice(K1{}, F{}, T{}, rs_st); slice(K2{}, F{}, T{}, rs_st); asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
      else { slice(K1{}, F{}, F{}, rsNull); slice(K2{}, F{}, F{}, rsNull); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      OW_FENCE();
      slice(K3{}, F{}, F{}, rsNull);
      ++gs;
    }
    for (int s = 1; s < nk; ++s, ++gs) {
      slice(K0{}, F{}, F{}, rsNull); cursor_next();
      slice(K1{}, F{}, F{}, rsNull);
      slice(K2{}, F{}, F{}, rsNull);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      OW_FENCE();
      slice(K3{}, F{}, F{}, rsNull);
    }
    unsigned long long t_kend = 0;
    if (p.stamps && tid == 0 && ti < 16) t_kend = __builtin_amdgcn_s_memrealtime();
    // ---- tile end: accumulators -> packed fp16, one v_permlane32_swap per register pair so that a lane holds 8 consecutive columns ----
    {
      int tm, tn;
      tile_mn(first + ti * nblk, tm, tn);
      st_tile = (unsigned)(((long long)tm * 256 * p.N + tn * 256) * 2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
        unsigned q[4][2];                                            // group g (columns 8 g + 4 kh ..+3): two packed registers
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            q[g][e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{acc[i][j][4 * g + 2 * e], acc[i][j][4 * g + 2 * e + 1]}, f16x2));
        // swap(X = group 2h, Y = group 2h + 1): the kh = 0 lane ends with [its group 2h | the partner's group 2h] = columns 16 h .. 16 h + 7,
        // the kh = 1 lane with [the partner's group 2h + 1 | its own] = columns 16 h + 8 .. 16 h + 15
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned o[4];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(q[2 * h][e], q[2 * h + 1][e], false, false);
            o[e] = sw[0]; o[2 + e] = sw[1];
          }
          pk[i][j][h] = u32x4{o[0], o[1], o[2], o[3]};
        }
      }
    if (!OVERLAP || ti + 1 == n_my) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { store_blk(i, j, 0); store_blk(i, j, 1); }
    }
    if (p.stamps && tid == 0 && ti < 16) {
      unsigned long long* sp = p.stamps + ((long long)blockIdx.x * 16 + ti) * 3;
      sp[0] = t_begin; sp[1] = t_kend; sp[2] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// =====================================================================================================================================
// v3 (late round 5): the v1 K loop on v_mfma_f32_16x16x32_f16 instead of 32x32x16.  MI355X_MICROARCH.md (DVFS item 7): on random operands the
// 16x16x32 loop delivers ~1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles (the chip holds a higher clock), and the shipped
// ping-pong kernel as well as hipBLASLt's assembly kernel (profiles/r05/gemm_vs_hipblaslt.txt) use that shape.  Same LDS image, same DMA;
// a K step = two 32-deep slices of 64 MFMAs (8 x 8 accumulator tiles of 16 x 16, 256 registers); the fragments of the next slice (16
// ds_read_b128) are read behind MFMAs 0..15, the 16 DMA instructions of K step s + 2 behind every third MFMA of slice (s, 1); step s is
// closed (lgkmcnt(0), vmcnt(0), barrier) after slice (s, 0), when all its fragments are in registers.
// =====================================================================================================================================
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int VARIANT>
__global__ void __launch_bounds__(256, 1)
onewave16_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, _Float16* __restrict__ C, int M, int N, int K,
                 unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[OW_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = N / 256;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int nk = K / 64;
  const int dr = lane >> 3, slot = lane & 7;
  unsigned voA[8], voB[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int row = 64 * wave + 8 * g + dr;
    const int c = slot ^ ((row >> 1) & 7);
    voA[g] = (unsigned)(((long long)(tm * 256 + row) * K) * 2 + c * 16);
    voB[g] = (unsigned)(((long long)(tn * 256 + row) * K) * 2 + c * 16);
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Bb = reinterpret_cast<const char*>(B);
  unsigned char* dstA = lds + wave * 8192;
  unsigned char* dstB = lds + OW_HT + wave * 8192;
  auto dma_a = [&](int step, int g) { OW_GLDS(Ab + voA[g] + step * 128, dstA + (step & 1) * OW_BUF + g * 1024); };
  auto dma_b = [&](int step, int g) { OW_GLDS(Bb + voB[g] + step * 128, dstB + (step & 1) * OW_BUF + g * 1024); };
  // fragment of mfma_f32_16x16x32_f16: per lane 8 consecutive k of row (lane & 15): k = 32 ks + 8 (lane >> 4)
  const int r16 = lane & 15, kq = lane >> 4;
  auto frag = [&](int buf, int operand, int rb, int ks) {
    const int row = 128 * (operand ? wc : wr) + 16 * rb + r16;
    const int c = (4 * ks + kq) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const f16x8*>(lds + buf * OW_BUF + operand * OW_HT + row * 128 + c * 16);
  };
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 8; ++g) { dma_a(0, g); dma_b(0, g); }
  if (nk > 1) {
#pragma unroll
    for (int g = 0; g < 8; ++g) { dma_a(1, g); dma_b(1, g); }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  OW_FENCE();
  unsigned long long t0 = 0;
  if (stamps && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
  f16x8 af[2][8], bf[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { af[0][i] = frag(0, 0, i, 0); bf[0][i] = frag(0, 1, i, 0); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  OW_FENCE();
  for (int s = 0; s < nk; ++s) {
    const int buf = VARIANT == 1 ? 0 : (s & 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cur = ks, nxt = ks ^ 1;
      const bool more = ks < 1 || s + 1 < nk;
      const int nbuf = ks < 1 ? buf : (VARIANT == 1 ? 0 : (buf ^ 1)), nks = ks ^ 1;
      const bool dma_on = VARIANT != 1 && ks == 1 && s + 2 < nk;
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        const int i = t >> 3, j = t & 7;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        if (t < 16) {
          if (more) {
            if (t < 8) af[nxt][t] = frag(nbuf, 0, t, nks);
            else bf[nxt][t - 8] = frag(nbuf, 1, t - 8, nks);
          }
        } else if (dma_on && (t - 16) % 3 == 0) {
          const int u = (t - 16) / 3, g = u >> 1;                     // u = 0 .. 15
          if ((u & 1) == 0) dma_a(s + 2, g); else dma_b(s + 2, g);
        }
        OW_FENCE();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ks == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      OW_FENCE();
    }
  }
  if (stamps && tid == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
  // epilogue (plain): lane holds column n = 16 j + r16, rows m = 16 i + 4 kq + r
  _Float16* cp = C + (long long)(tm * 256 + 128 * wr) * N + tn * 256 + 128 * wc;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cp[(long long)(16 * i + 4 * kq + r) * N + 16 * j + r16] = (_Float16)acc[i][j][r];
}

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void fill_f16(_Float16* p, long long n, unsigned seed, int kind) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned h = hash32((unsigned)i * 2654435761U + seed);
    p[i] = kind == 1 ? (_Float16)(float)((int)(h % 3u) - 1) : (_Float16)((h >> 8) * (1.0f / 8388608.0f) - 1.0f);
  }
}
__global__ void ref_check(const _Float16* A, const _Float16* B, const _Float16* C, int M, int N, int K, unsigned long long* nbad, int stride) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * stride;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf((float)A[(long long)m * K + k], (float)B[(long long)n * K + k], s);
  if ((float)C[i] != (float)(_Float16)s) atomicAdd(nbad, 1ull);
}

template <int V, bool BUF = false>
static void launch(const _Float16* A, const _Float16* B, _Float16* C, int M, int N, int K, unsigned long long* st, hipStream_t s) {
  hipLaunchKernelGGL((onewave_kernel<V, BUF>), dim3((M / 256) * (N / 256)), dim3(256), 0, s, A, B, C, M, N, K, st);
}

int main() {
  hipStream_t s; HIP_OK(hipStreamCreate(&s));
  struct Shape { int M, N, K; const char* name; };
  const Shape shapes[] = {{4096, 4096, 4096, "4k"}, {98304, 2304, 768, "qkv (384 M tiles)"}, {98304, 768, 3072, "c_proj"}, {8192, 8192, 8192, "8k"}};
  unsigned long long *d_bad, *d_st; HIP_OK(hipMalloc(&d_bad, 8)); HIP_OK(hipMalloc(&d_st, 16 * 65536));
  for (const Shape& sh : shapes) {
    const long long MK = (long long)sh.M * sh.K, NK = (long long)sh.N * sh.K, MN = (long long)sh.M * sh.N;
    _Float16 *A, *B, *C, *C2; HIP_OK(hipMalloc(&A, MK * 2)); HIP_OK(hipMalloc(&B, NK * 2)); HIP_OK(hipMalloc(&C, MN * 2)); HIP_OK(hipMalloc(&C2, MN * 2));
    // correctness on small integers (exact in any summation order)
    fill_f16<<<2048, 256, 0, s>>>(A, MK, 11u, 1); fill_f16<<<2048, 256, 0, s>>>(B, NK, 13u, 1);
    HIP_OK(hipMemsetAsync(C, 0xff, MN * 2, s)); HIP_OK(hipMemsetAsync(d_bad, 0, 8, s));
    launch<0>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
    const int stride = MN > (1ll << 24) ? 97 : 1;
    ref_check<<<(unsigned)((MN / stride + 255) / 256), 256, 0, s>>>(A, B, C, sh.M, sh.N, sh.K, d_bad, stride);
    unsigned long long bad; HIP_OK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
    printf("check %-18s M=%d N=%d K=%d: %llu wrong of %lld sampled %s\n", sh.name, sh.M, sh.N, sh.K, bad, MN / stride, bad ? "FAIL" : "OK");
    HIP_OK(hipMemsetAsync(C, 0xff, MN * 2, s)); HIP_OK(hipMemsetAsync(d_bad, 0, 8, s));
    launch<0, true>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
    ref_check<<<(unsigned)((MN / stride + 255) / 256), 256, 0, s>>>(A, B, C, sh.M, sh.N, sh.K, d_bad, stride);
    HIP_OK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
    printf("check %-18s (buffer DMA): %llu wrong %s\n", sh.name, bad, bad ? "FAIL" : "OK");
    // timing on uniform random operands, interleaved rounds
    fill_f16<<<2048, 256, 0, s>>>(A, MK, 3u, 0); fill_f16<<<2048, 256, 0, s>>>(B, NK, 5u, 0);
    const double flop = 2.0 * sh.M * sh.N * sh.K;
    std::vector<float> t[5];
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (int round = 0; round < 12; ++round)
      for (int v = 0; v < 5; ++v) {
        HIP_OK(hipEventRecord(e0, s));
        if (v == 4) launch<0, true>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
        else if (v == 0) launch<0>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
        else if (v == 1) launch<1>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
        else if (v == 2) launch<2>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
        else OVIS_OKAY(ovis_gemm_nt_f16(A, sh.K, B, sh.K, C2, sh.N, sh.M, sh.N, sh.K, nullptr, nullptr, 0, 0, 1, s));
        HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (round >= 2) t[v].push_back(ms);
      }
    const char* names[5] = {"one wave / SIMD, reads + DMA between MFMAs", "  ... without the DMA (step-0 operands)", "  ... without the MFMAs (DMA + reads + barriers)",
                            "shipped ping-pong kernel (ovis_gemm_nt_f16)", "one wave / SIMD, DMA = buffer_load lds (SGPR K offset)"};
    for (int v = 0; v < 5; ++v) {
      std::sort(t[v].begin(), t[v].end());
      const float med = t[v][t[v].size() / 2], mn = t[v][0];
      printf("time  %-18s %-50s median %.4f ms (%4.0f TF)  min %.4f ms (%4.0f TF)\n", sh.name, names[v], med, flop / med / 1e9, mn, flop / mn / 1e9);
    }
    // ---- v3: the one-wave K loop on 16x16x32 MFMAs ----
    {
      fill_f16<<<2048, 256, 0, s>>>(A, MK, 11u, 1); fill_f16<<<2048, 256, 0, s>>>(B, NK, 13u, 1);
      HIP_OK(hipMemsetAsync(C, 0xff, MN * 2, s)); HIP_OK(hipMemsetAsync(d_bad, 0, 8, s));
      hipLaunchKernelGGL((onewave16_kernel<0>), dim3((sh.M / 256) * (sh.N / 256)), dim3(256), 0, s, A, B, C, sh.M, sh.N, sh.K, (unsigned long long*)nullptr);
      ref_check<<<(unsigned)((MN / stride + 255) / 256), 256, 0, s>>>(A, B, C, sh.M, sh.N, sh.K, d_bad, stride);
      HIP_OK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
      printf("check %-18s one-wave kernel on 16x16x32 MFMAs: %llu wrong %s\n", sh.name, bad, bad ? "FAIL" : "OK");
      fill_f16<<<2048, 256, 0, s>>>(A, MK, 3u, 0); fill_f16<<<2048, 256, 0, s>>>(B, NK, 5u, 0);
      std::vector<float> t3[4];
      for (int round = 0; round < 12; ++round)
        for (int v = 0; v < 4; ++v) {
          HIP_OK(hipEventRecord(e0, s));
          if (v == 0) hipLaunchKernelGGL((onewave16_kernel<0>), dim3((sh.M / 256) * (sh.N / 256)), dim3(256), 0, s, A, B, C, sh.M, sh.N, sh.K, (unsigned long long*)nullptr);
          else if (v == 1) hipLaunchKernelGGL((onewave16_kernel<1>), dim3((sh.M / 256) * (sh.N / 256)), dim3(256), 0, s, A, B, C, sh.M, sh.N, sh.K, (unsigned long long*)nullptr);
          else if (v == 2) launch<0>(A, B, C, sh.M, sh.N, sh.K, nullptr, s);
          else OVIS_OKAY(ovis_gemm_nt_f16(A, sh.K, B, sh.K, C2, sh.N, sh.M, sh.N, sh.K, nullptr, nullptr, 0, 0, 1, s));
          HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
          float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
          if (round >= 2) t3[v].push_back(ms);
        }
      const char* n3[4] = {"one wave / SIMD on 16x16x32 MFMAs", "  ... without the DMA (step-0 operands)", "one wave / SIMD on 32x32x16 MFMAs (v1)", "shipped ping-pong kernel (ovis_gemm_nt_f16)"};
      for (int v = 0; v < 4; ++v) {
        std::sort(t3[v].begin(), t3[v].end());
        const float med = t3[v][t3[v].size() / 2], mn = t3[v][0];
        printf("time3 %-18s %-50s median %.4f ms (%4.0f TF)  min %.4f ms (%4.0f TF)\n", sh.name, n3[v], med, flop / med / 1e9, mn, flop / mn / 1e9);
      }
      const int nt3 = (sh.M / 256) * (sh.N / 256);
      if (nt3 <= 65536) {
        hipLaunchKernelGGL((onewave16_kernel<0>), dim3(nt3), dim3(256), 0, s, A, B, C, sh.M, sh.N, sh.K, d_st);
        std::vector<unsigned long long> st(2 * nt3);
        HIP_OK(hipMemcpyAsync(st.data(), d_st, 16ull * nt3, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
        std::vector<double> span;
        for (int i = 0; i < nt3; ++i) span.push_back((st[2 * i + 1] - st[2 * i]) * 0.01 / (sh.K / 64));
        std::sort(span.begin(), span.end());
        printf("trace3 %-17s K loop per 64-deep step on 16x16x32 MFMAs (in-kernel, median over %d tiles): %.3f us (p10 %.3f, p90 %.3f)\n", sh.name, nt3, span[nt3 / 2], span[nt3 / 10], span[nt3 * 9 / 10]);
      }
    }
    if (getenv("OW_ONLY_V3")) { HIP_OK(hipFree(A)); HIP_OK(hipFree(B)); HIP_OK(hipFree(C)); HIP_OK(hipFree(C2)); continue; }
    // ---- v2: persistent, overlapped epilogue ----
    {
      OWArgs a; a.A = A; a.B = B; a.C = C; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.tiles_m = sh.M / 256; a.tiles_n = sh.N / 256; a.n_tiles = a.tiles_m * a.tiles_n;
      const int groups = (a.tiles_n + 5) / 6;                        // column groups of <= 6 N tiles (the shipped kernel's raster)
      a.grp_w = a.tiles_n / groups; a.grp_rem = a.tiles_n % groups; a.stamps = nullptr;
      const int grid = std::min(256, a.n_tiles);
      // correctness (small integers)
      fill_f16<<<2048, 256, 0, s>>>(A, MK, 11u, 1); fill_f16<<<2048, 256, 0, s>>>(B, NK, 13u, 1);
      for (int ov = 0; ov < 2; ++ov) {
        HIP_OK(hipMemsetAsync(C, 0xff, MN * 2, s)); HIP_OK(hipMemsetAsync(d_bad, 0, 8, s));
        if (ov) hipLaunchKernelGGL((onewave_persist_kernel<true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((onewave_persist_kernel<false>), dim3(grid), dim3(256), 0, s, a);
        ref_check<<<(unsigned)((MN / stride + 255) / 256), 256, 0, s>>>(A, B, C, sh.M, sh.N, sh.K, d_bad, stride);
        HIP_OK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
        printf("check %-18s persistent one-wave kernel (stores %s): %llu wrong %s\n", sh.name, ov ? "between the next tile's MFMAs" : "behind the conversion", bad, bad ? "FAIL" : "OK");
      }
      fill_f16<<<2048, 256, 0, s>>>(A, MK, 3u, 0); fill_f16<<<2048, 256, 0, s>>>(B, NK, 5u, 0);
      std::vector<float> tp[3];
      for (int round = 0; round < 14; ++round)
        for (int v = 0; v < 3; ++v) {
          HIP_OK(hipEventRecord(e0, s));
          if (v == 0) hipLaunchKernelGGL((onewave_persist_kernel<false>), dim3(grid), dim3(256), 0, s, a);
          else if (v == 1) hipLaunchKernelGGL((onewave_persist_kernel<true>), dim3(grid), dim3(256), 0, s, a);
          else OVIS_OKAY(ovis_gemm_nt_f16(A, sh.K, B, sh.K, C2, sh.N, sh.M, sh.N, sh.K, nullptr, nullptr, 0, 0, 1, s));
          HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
          float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
          if (round >= 2) tp[v].push_back(ms);
        }
      const char* pn[3] = {"PERSISTENT one wave / SIMD, stores behind the conversion", "PERSISTENT one wave / SIMD, stores between the next tile's MFMAs",
                           "shipped ping-pong kernel (ovis_gemm_nt_f16)"};
      for (int v = 0; v < 3; ++v) {
        std::sort(tp[v].begin(), tp[v].end());
        const float med = tp[v][tp[v].size() / 2], mn = tp[v][0];
        printf("time2 %-18s %-66s median %.4f ms (%4.0f TF)  min %.4f ms (%4.0f TF)\n", sh.name, pn[v], med, flop / med / 1e9, mn, flop / mn / 1e9);
      }
      // per-tile spans (100 MHz ticks) of the overlapped variant
      a.stamps = d_st;
      HIP_OK(hipMemsetAsync(d_st, 0, 256 * 16 * 3 * 8, s));
      hipLaunchKernelGGL((onewave_persist_kernel<true>), dim3(grid), dim3(256), 0, s, a);
      std::vector<unsigned long long> st(256 * 16 * 3);
      HIP_OK(hipMemcpyAsync(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
      const int per = std::min(16, a.n_tiles / grid);
      for (int ti = 0; ti < std::min(per, 4); ++ti) {
        std::vector<double> kl, cv, gap;
        for (int w = 0; w < grid; ++w) {
          const unsigned long long* sp = &st[(w * 16 + ti) * 3];
          if (!sp[0]) continue;
          kl.push_back((sp[1] - sp[0]) * 0.01); cv.push_back((sp[2] - sp[1]) * 0.01);
          if (ti + 1 < per && st[(w * 16 + ti + 1) * 3]) gap.push_back((st[(w * 16 + ti + 1) * 3] - sp[0]) * 0.01);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("trace2 %-17s tile %d: K loop %.2f us (%.3f per K step), conversion %.2f us, tile period %.2f us\n", sh.name, ti, med(kl), med(kl) / (sh.K / 64), med(cv), med(gap));
      }
    }
    // in-kernel span of the K loop (100 MHz ticks), variant 0
    const int nt = (sh.M / 256) * (sh.N / 256);
    if (nt <= 65536) {
      launch<0>(A, B, C, sh.M, sh.N, sh.K, d_st, s);
      std::vector<unsigned long long> st(2 * nt);
      HIP_OK(hipMemcpyAsync(st.data(), d_st, 16ull * nt, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
      std::vector<double> span;
      for (int i = 0; i < nt; ++i) span.push_back((st[2 * i + 1] - st[2 * i]) * 0.01 / (sh.K / 64));
      std::sort(span.begin(), span.end());
      printf("trace %-18s K loop per 64-deep step (in-kernel, median over %d tiles): %.3f us (p10 %.3f, p90 %.3f)  [ping-pong kernel: 1.59-1.66 us, profiles/r03/lab_trace_qkv.txt]\n",
             sh.name, nt, span[nt / 2], span[nt / 10], span[nt * 9 / 10]);
    }
    HIP_OK(hipFree(A)); HIP_OK(hipFree(B)); HIP_OK(hipFree(C)); HIP_OK(hipFree(C2));
  }
  printf("gemm_onewave_lab done\n");
  return 0;
}
