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
//   * the stream of K steps runs across output tiles: BOTH K steps of the next tile are in flight before the epilogue of the
//     current one starts (its stores sit behind them in the in-order vmcnt queue and get two K steps to drain), and the
//     waits of those two K steps count the stores as "younger" instead of waiting for them;
//   * tile transition: G0 waits one slot for G1's last MFMAs, then all 8 wavefronts convert + store in the same slot (a
//     wavefront issues a 1-KB store every ~100 ns whatever the others do; two groups storing in turn cost twice the time);
//     row block outer / column pair inner store order, so that the two 64-byte halves of a 128-byte line are written by
//     consecutive instructions (5-6 % on the whole GEMM against the other order: partial-line writes);
//   * accumulators start at the bias (+ the f32 residual) instead of zero: the epilogue only activates, converts, stores;
//   * edge tiles are shifted inside the matrix (loads) and masked (stores): DMA addresses = lane constant + tile base;
//     the workgroup's tile list (bases, owned / computed rows and columns) is built once in LDS;
//   * LDS rows are 128 B (a full cache line per DMA'd row segment); the 16-byte chunk index is XOR-swizzled with
//     (row>>1)&7 on the DMA's per-lane SOURCE address and on the fragment read (conflict-free ds_read_b128);
//   * the weight rows of a wavefront's N range are DMA'd in a permuted order (n = 32j + 8q + 4e + r for MFMA row 4q + r of
//     tile 2j + e), so that a lane ends up with 8 consecutive output columns: one 16-byte store per fp16 row segment;
//   * bias lives in LDS for the whole launch (read with ds_read, so it never touches the vmcnt queue).
// Measured (tools/gemm_lab.cpp, interleaved with gemm_f16_256_kernel in one process, M = 98 500): QKV 999 vs 889 TF, fc1 1013 vs
// 860, out-proj 549 vs 480, fc2 991 vs 851; 4096^3 1291-1350 vs 1135-1145.  The K loop itself runs at 1.43 us per 64-deep step
// (~1390 TF); what is left on the K = 768 shapes is the per-tile epilogue (activation + conversion VALU, 16-32 stores per
// wavefront) and, for the residual GEMMs, the 128 MB burst of residual loads + C stores every round of tiles.
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
#include <mutex>
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int PP_HT = 128 * 128;           // bytes of a half-tile: 128 rows x 64 halfs
constexpr int PP_BUF = 4 * PP_HT;          // one K step: A0 A1 B0 B1
constexpr int PP_BIAS = 2 * PP_BUF;        // bias (f32) behind the two buffers
constexpr int PP_LDS = 160 * 1024;
constexpr int PP_TAB = PP_LDS - 8192;      // this workgroup's tile list: 256 entries of 32 bytes (1000 crops x ViT-L/14@336 fc1: 141 per workgroup)
constexpr int PP_MAX_TILES = 256;
constexpr int PP_MAX_BIAS_N = (PP_TAB - PP_BIAS) / 4;
constexpr int PP_FLAG = PP_BIAS + 16384;   // FH: one word behind 16 KB of bias (N <= 4096 there): "a result of this workgroup was not finite"
// LNF (LayerNorm folded into the GEMM): the tile list is capped at 128 entries (4 KB) and the upper 4 KB of the list region hold two
// 2 KB slots of per-row statistics (mean, rstd of the tile's 256 rows), filled by LDS-DMA one tile ahead
constexpr int PP_STATS = PP_LDS - 4096;
constexpr int PP_MAX_TILES_LNF = 128;
struct PPTile { long long a_off, b_off; int bm, bml, bn, bnl; };   // DMA bases (bytes) of the shifted tile; rows / columns owned and computed

struct PPArgs {
  const _Float16* A; const _Float16* B; void* C; const float* bias; const float* R;
  long long lda, ldb, ldc, ldr;
  long long planeA, planeB, planeC;          // X3: bytes between the bf16 planes of A / B (and of C when the output is planes)
  int M, N, K, act;
  int tiles_m, tiles_n, n_tiles;
  int grp_w, grp_rem;                       // raster: column groups of grp_w (+1 for the first grp_rem groups) N tiles
  int desync_ns;                            // start offset spread over the workgroups that own one tile fewer (ns)
  int dbg;                                  // lab only: 8 = start offsets per XCD instead of per workgroup, 16 = no s_setprio for G1's epilogue
                                            // (round 3's flags 2 / 4 = operands of tile (0,0) and 32 = no MFMAs are gone: profiles/r03/lab_l2_locality.txt, lab_skeleton.txt)
                                            // (the store-suppressing flags 1 / 2 / 32 of the round-2 experiments are gone: profiles/r02/lab_ub*.txt)
  unsigned long long* stamps;               // lab only: s_memrealtime stamps [workgroup][tile iteration < 16][2 groups][4]
  void* dump;                               // 2 KB scratch that the masked lanes of edge tiles store to (never read)
  const float* ln_stats = nullptr;          // LNF: [M][2] (mean, rstd) of the rows of A
  const float* ln_s = nullptr;              // LNF: [N] row sums of the gamma-folded weight (f32)
  float* ln_part = nullptr;                 // PSTAT: [M][ln_pslots][2] partial (sum, sum of squares) of the fp16 output rows, one slot per
  int ln_pslots = 0;                        //        (N tile, wavefront column): slot = 4 * (N tile) + wc
  const float* lno_gamma = nullptr;         // LNO: LayerNorm weight / bias [N] applied to the output rows (N == 256: a tile holds whole rows)
  const float* lno_beta = nullptr;
  float lno_eps = 0.f;
  int cv_H = 0, cv_W = 0, cv_C = 0;         // CV: 3x3 / stride 1 / pad 1 convolution; A = the ZERO-PADDED input [T][cv_H + 2][cv_W + 2][cv_C] (f32),
                                            //     M = T cv_H cv_W output pixels, K = 9 cv_C
  float a_scale = 1.f, acc_scale = 1.f, out_scale = 1.f;   // FH: A is multiplied by a_scale while split; accumulators live at scale a_scale * w_scale
  int* range_flag = nullptr;                // FH: set to 1 when a tile's result is not finite (an operand left the fp16 range)
  void* C2 = nullptr; long long ldc2 = 0;   // DUAL: the columns [r_col0, N) go to C2 [M, N - r_col0] (row stride ldc2), the columns [0, r_col0) to C
  int r_mod = 0, r_col0 = 0;                // DUAL: ... and get R[m % r_mod][n - r_col0] added (R: r_mod rows of N - r_col0 columns, row stride ldr)
  int dual_T = 0;                           // DUAL: M / r_mod (rows are tiled per period)
};

#define PP_GLDS(src, dst) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)

// OUT: 0 f32, 1 fp16, 2 three bf16 planes (the exact split of the f32 result, operand format of the X3 mode).
// X3 : f32-grade GEMM on the bf16 matrix cores (gemm_f32x3.h): A and B arrive as the exact 3-way bf16 split of the f32 operands
//      (planes [3][rows][ld]); the K loop runs over the six plane pairs (a2 b0, a0 b2, a1 b1, a1 b0, a0 b1, a0 b0: smallest terms
//      first) x K, i.e. this is the same kernel on a K axis of 6 K -- the split costs no VALU work in the loop.  Accumulators start
//      at zero and bias / residual are added in the epilogue (an O(1) start value would cost the f32-grade error bound).
// FA : "f32 A" mode (round 2; bf16x2 of gemm_f32x3.h on this schedule): A is the f32 activation matrix itself, DMA'd as f32 (a K
//      step is 32 floats = the same 128-byte LDS rows, same swizzle); B are the two leading bf16 planes of the constant weight
//      (64-byte rows, 16 rows per DMA instruction, plane 0 / plane 1 of a wavefront's 16 rows side by side: a fragment read is a
//      linear 1 KB -> conflict-free without a swizzle).  A wavefront reads its f32 fragment (8 consecutive k per lane), splits it
//      into hi / lo bf16 in registers (3 VALU per element) and issues the three products a1 b0, a0 b1, a0 b0 -- no split pass over
//      A in memory, no plane stores to LDS, and the DMA pipeline / barrier protocol / vmcnt accounting of the fp16 kernel unchanged
//      (every half-tile is still two DMA instructions per wavefront).
// R16: the residual is fp16 (and the output too, OUT == 1): the fp16 residual stream of the CLIP tower -- what the reference's
//      fp16 CLIP keeps between blocks (adapter.py:108-111); a lane's 8 consecutive columns of a row are ONE 16-byte load.
// TM : rows of a tile, 256 or 192.  192 = the same schedule with 96 rows per wave group (row-block quadrant i = 1 has two 16-row
//      blocks instead of four: 3/4 of the MFMAs and stores, the DMA volume of a 256-row tile); chosen per launch when it saves a
//      whole round of tiles (e.g. N = K = 256 at M = 96 600: 378 tiles = 1.48 rounds of 256 workgroups -> 504 tiles = 1.97).
// EPI (fp16 output only): bit 0 = full-line stores -- the lanes l and l ^ 8 of a 16-lane row exchange one 16-byte pack (DPP row_ror:8),
//      so that a store instruction writes 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B; bits 1-2 = cache policy of the
//      C stores: 0 default, 1 sc1 (write-through: the line is not kept in the XCD's L2, which the operand panels need), 2 nt.
// LNF: C = act(LayerNorm(A) W^T + b) computed from the raw rows of A: with W' = gamma * W (folded once), s[n] = sum_k W'[n,k] and
//      c[n] = b[n] + sum_k beta[k] W[n,k] it is rstd[m] * (A W'^T - mean[m] s[n]) + c[n].  The accumulators start at -mean[m] s[n], the
//      epilogue applies rstd[m] and c[n] (`bias` = c).  mean / rstd of a tile's 256 rows come from LDS: every wavefront fetches 32 rows
//      of the NEXT-BUT-ONE tile's statistics with one 4-byte-per-lane LDS-DMA right after the tile transition -- one more operation in
//      the in-order vmcnt queue exactly where the epilogue's stores sit, so the counted waits of the K loop only grow by one.
// PSTAT (fp16 residual GEMMs): the epilogue also emits, per output row and per (N tile, wavefront column), the sum and the sum of squares
//      of the 64 fp16 values it stores -- the LayerNorm statistics of the NEXT block's ln are then a 12-term sum per row
//      (row_stats_finalize_kernel) instead of a pass over the 151 MB residual stream.  Deterministic: fixed slots, fixed order.
// LNO (f32 output, N == 256): C = LayerNorm(A W^T + b + R) -- the post-norm of the pixel decoder's encoder layers (msdeformattn.py:139-146
//      after output_proj / linear2): a 256-column tile holds whole rows, so the epilogue normalises them before they are stored and
//      the LayerNorm kernel's pass over the [M, 256] tensor (99 MB in, 99 MB out per launch at 720p) disappears.  Row sums of the four
//      wavefront columns meet in LDS (8 KB of the bias region) across one extra workgroup barrier inside the epilogue -- both wave groups
//      run their epilogue in the same slot, so the barrier counts of the two groups stay equal.  Single pass (E[x^2] - mean^2) in f32.
// FH (f32-A mode, round 4: "fp16x2"): the same three products on the FP16 matrix cores.  a = hi + lo with hi = fp16(a), lo = fp16(a - hi):
//      11 + 11 significand bits per operand (bf16x2: 8 + 8), so hi hi + hi lo + lo hi carries every term down to 2^-22 |a b| -- the f32
//      grade -- at the MFMA cost of bf16x2.  What fp16 lacks is range (max 65 504, subnormal spacing 2^-24), so both operands are moved
//      to the top of it by powers of two: the constant weight planes are stored as fp16 hi / lo of w * w_scale (w_scale = 2^k with
//      max |w| w_scale in [2^14, 2^15), chosen when the planes are built), the activations are multiplied by a_scale (a launch
//      parameter, default 16: |a| < 4 094, absolute floor of lo 2^-29) while they are split in registers.  The accumulators then live
//      at scale a_scale * w_scale; they start at ZERO and the epilogue multiplies by the inverse scale (exact) and adds bias / residual, as
//      in the X3 mode: the 16-bit MFMA aligns its 32 products to the accumulator and truncates, so an O(1) start value cost 1.2e-6 of the
//      row scale on rows whose products are small against bias + residual, 5 x the f32 MFMA kernel (profiles/r04/fp16x2_error.txt).  An
//      activation beyond the range becomes inf, the tile's result inf / NaN: the epilogue tests its (scaled) accumulators with an
//      fma-by-zero chain and raises p.range_flag through an LDS word (no extra vector-memory operation inside the counted waits);
//      callers then repeat the work under bf16x3 (openvis_amd/modeling/video_maskformer.py).
template <int OUT, int ACT, bool HAS_R, bool X3, bool FA = false, bool R16 = false, int TM = 256, int EPI = 0, bool LNF = false, bool PSTAT = false,
          bool LNO = false, bool CV = false, bool FH = false, bool DUAL = false>
__global__ void __launch_bounds__(512)
gemm_f16_pp_kernel(const PPArgs p) {
  static_assert(!FH || FA, "FH: the fp16 split of the f32-A mode");
  // DUAL (fp16x2, round 4): one GEMM, two outputs, a row-periodic residual on the second -- the value projection and the fused
  // sampling-offset / attention-weight projection of a deformable-attention encoder layer (ms_deform_attn.py:98-104) read the same rows:
  // value = src Wv^T + bv, oa = (src + pos) Woa^T + boa = src Woa^T + (pos Woa^T) + boa with pos identical for every frame.  W = [Wv ; Woa]
  // ([N, K], N = 256 + 288), the columns below r_col0 (a multiple of 256: no tile straddles it) are stored to C, the others to C2 after
  // R[m % r_mod] (= pos Woa^T, computed once per shape) is added.  src is read once instead of three times (add, value GEMM, oa GEMM) and
  // the 288-column GEMM, which no 256-column tiling fits, disappears into a launch whose tail tile is shifted like any edge tile.
  static_assert(!DUAL || (FH && HAS_R && OUT == 0 && ACT == 0 && !LNO && !CV), "DUAL: fp16x2, f32 outputs, residual added in the epilogue");
  // CV (f32-A mode): implicit GEMM of a 3x3 / stride 1 / pad 1 convolution over a zero-padded NHWC input.  Row m of the GEMM is output
  // pixel (t, y, x); K step k covers 32 channels of tap (kh, kw) = k / (C / 32): the row's 128 bytes sit at
  //   rowbase(m) + ((kh (W + 2) + kw) C + 32 (k % (C / 32))) 4,     rowbase(m) = ((t (H + 2) + y) (W + 2) + x) C 4
  // -- a per-lane row base (recomputed when a DMA cursor enters a new tile: two integer divisions per row, after a phase's MFMAs) plus a
  // wave-uniform offset per K step.  Nothing else changes: same DMA count, same waits, same epilogue (C is [M, N], rows = pixels).
  static_assert(!CV || (FA && !X3), "CV: the f32-A (bf16x2) mode");
  static_assert(!LNO || (OUT == 0 && !X3 && ACT == 0 && EPI == 0), "LNO: f32 output, no activation");
  static_assert(!PSTAT || (R16 && (EPI & 1)), "PSTAT: fp16 residual GEMM with the full-line epilogue");
  static_assert(!LNF || (OUT == 1 && !HAS_R && !X3 && !FA && TM == 256 && (EPI & 1)), "LNF: fp16 output, full-line epilogue, 256-row tiles");
  static_assert(EPI == 0 || (OUT == 1 && !X3 && (!HAS_R || R16)), "EPI variants: fp16 output (fp16 residual or none)");
  static_assert(TM == 256 || (TM == 192 && !X3), "tile heights: 256, or 192 for the fp16 / f32-A modes");
  constexpr int GS = TM / 2;                                       // rows of a wave group
  // TM = 192: the second half-tile of a group starts 32 rows into the group (rows GS wr + 32 .. + 95, overlapping the first by 32 rows)
  // and only its upper two 16-row blocks are multiplied -- no load reaches beyond the tile's 192 rows
  constexpr int A1_ROW = TM == 256 ? 64 : 32;                      // first row (inside the group) of half-tile i = 1
  constexpr int MB1_LO = TM == 256 ? 0 : 2;                        // first 16-row block of half-tile i = 1 that is used
  constexpr int MBT = 8 - MB1_LO;                                  // 16-row blocks of a wavefront (accumulator rows 0 .. MBT-1)
  static_assert(!(FA && X3) && (!FA || OUT == 0), "f32-A mode: f32 output, no plane walking");
  static_assert(!R16 || (HAS_R && OUT == 1 && !X3 && !FA), "fp16 residual: fp16 output of the fp16 GEMM");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS];   // ONE LDS object (a second one de-pipelines the DMA)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, q = lane >> 4, sw = l15 >> 1;
  const int nblk = gridDim.x;
  const int nk1 = FA ? p.K >> 5 : p.K >> 6;                        // K steps of one plane pair (FA: 32 floats per step)
  const int nk = X3 ? 6 * nk1 : nk1;

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

  // bias (zeros when there is none: no branch in the epilogue) and this workgroup's tile list go to LDS once; a tile
  // change in the main loop is then one broadcast ds_read_b128 instead of two integer divisions
  const int first = (int)ovis::xcd_remap(blockIdx.x, nblk);
  const int n_my = (p.n_tiles - first + nblk - 1) / nblk;
  for (int i = tid * 4; i < p.N; i += 512 * 4)
    *reinterpret_cast<float4*>(lds + PP_BIAS + i * 4) =
        p.bias ? *reinterpret_cast<const float4*>(p.bias + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (LNF)
    for (int i = tid * 4; i < p.N; i += 512 * 4)
      *reinterpret_cast<float4*>(lds + PP_BIAS + (p.N + i) * 4) = *reinterpret_cast<const float4*>(p.ln_s + i);
  if constexpr (LNO)                                               // N == 256: bias [0,1 KB), gamma [1,2 KB), beta [2,3 KB), row partials [4,12 KB)
    for (int i = tid * 4; i < p.N; i += 512 * 4) {
      *reinterpret_cast<float4*>(lds + PP_BIAS + (p.N + i) * 4) = *reinterpret_cast<const float4*>(p.lno_gamma + i);
      *reinterpret_cast<float4*>(lds + PP_BIAS + (2 * p.N + i) * 4) = *reinterpret_cast<const float4*>(p.lno_beta + i);
    }
  if constexpr (FH) { if (tid == 0) *reinterpret_cast<int*>(lds + PP_FLAG) = 0; }
  for (int i = tid; i < n_my; i += 512) {
    int tm, tn;
    PPTile t;
    if constexpr (DUAL) {
      // rows are tiled PER FRAME (M = frames x r_mod; the last tile of a frame is shifted inside it) and enumerated row-tile major, frame,
      // column tile: the tiles that read the same rows of the periodic term R -- one per frame and second-output column tile -- are
      // consecutive, i.e. run on one XCD at the same time, and R comes from that XCD's L2 instead of once per frame from memory
      const int L = first + i * nblk, per = p.dual_T * p.tiles_n;
      const int rt = L / per, rem = L - rt * per, f = rem / p.tiles_n;
      tn = rem - f * p.tiles_n;
      t.bm = f * p.r_mod + rt * TM; t.bml = min(t.bm, (f + 1) * p.r_mod - TM);
      t.bn = tn * 256; t.bnl = min(t.bn, p.N - 256);
    } else {
    tile_mn(first + i * nblk, tm, tn);
    t.bm = tm * TM; t.bn = tn * 256; t.bml = min(t.bm, p.M - TM); t.bnl = min(t.bn, p.N - 256);
    }
    t.a_off = (long long)t.bml * p.lda * (FA ? 4 : 2); t.b_off = (long long)t.bnl * p.ldb * 2;
    *reinterpret_cast<PPTile*>(lds + PP_TAB + i * 32) = t;
  }
  __syncthreads();
  // wave-uniform copy of entry i.  Inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of an ordinary LDS load here
  // (LDS-DMA in flight may alias it for all the compiler knows), which drains the whole DMA pipeline at every tile change.
  auto tile_entry = [&](int i) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    u32x4 lo, hi;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_TAB) + i * 32;   // LDS byte address
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
    PPTile t;
    t.a_off = ((long long)__builtin_amdgcn_readfirstlane(lo.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(lo.x);
    t.b_off = ((long long)__builtin_amdgcn_readfirstlane(lo.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(lo.z);
    t.bm = __builtin_amdgcn_readfirstlane(hi.x); t.bml = __builtin_amdgcn_readfirstlane(hi.y);
    t.bn = __builtin_amdgcn_readfirstlane(hi.z); t.bnl = __builtin_amdgcn_readfirstlane(hi.w);
    return t;
  };

  // 8 consecutive bias values (columns col .. col + 7) from LDS.  Inline asm for the same reason as tile_entry: in front of an
  // ordinary LDS load of the bias hipcc waits for vmcnt(0) -- every store of the epilogue and both K steps of the next tile.
  auto bias8 = [&](int col, f32x4& b0, f32x4& b1) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_BIAS) + col * 4;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(addr) : "memory");
  };

  // LNF: (mean, rstd) of this lane's row in each of the 8 row blocks of its wave group, from statistics slot `slot` (one asm block:
  // eight reads in flight, one wait); and the LDS-DMA that fills a slot with the statistics of tile `i` (32 rows per wavefront)
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  auto stats8 = [&](int slot, f32x2 (&st)[8]) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_STATS) + slot * 2048 + (wr * GS + l15) * 8;
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:128\n\tds_read_b64 %2, %8 offset:256\n\tds_read_b64 %3, %8 offset:384\n\t"
                 "ds_read_b64 %4, %8 offset:512\n\tds_read_b64 %5, %8 offset:640\n\tds_read_b64 %6, %8 offset:768\n\tds_read_b64 %7, %8 offset:896\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(st[0]), "=&v"(st[1]), "=&v"(st[2]), "=&v"(st[3]), "=&v"(st[4]), "=&v"(st[5]), "=&v"(st[6]), "=&v"(st[7]) : "v"(addr) : "memory");
  };

  // ---- DMA sources: a lane-constant byte offset per (half-tile, 8-row group) + a wave-uniform tile base ----
  // Edge tiles are SHIFTED inside the matrix for the loads (rows min(256 t, M - 256) ...) and their stores are masked to
  // the rows / columns that belong to the tile: no per-lane clamping, so a tile change costs scalar arithmetic only.
  const int dr = lane >> 3;                                       // row inside the 8-row group
  unsigned voA[2][2], voB[2][2];                                  // [half-tile i / j][group g]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c = (lane & 7) ^ ((q + 4 * g) & 7);                 // logical chunk held by this lane's slot: ((row>>1)&7)
    const int i16 = 8 * g + dr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      voA[h][g] = (unsigned)((long long)(wr * GS + h * A1_ROW + 16 * (wave & 3) + i16) * p.lda * (FA ? 4 : 2) + c * 16);   // half-tile row 64 wr + 16 (wave&3) + i16
      if constexpr (FA) {     // instruction g = plane g; 16 rows x 64 B: lane -> (row lane>>2, 16-byte chunk lane&3)
        const int r16 = lane >> 2;
        voB[h][g] = (unsigned)(g * p.planeB + (long long)((wave >> 1) * 64 + 32 * h + 8 * (r16 >> 2) + 4 * (wave & 1) + (r16 & 3)) * p.ldb * 2 + (lane & 3) * 16);
      } else {
        voB[h][g] = (unsigned)((long long)((wave >> 1) * 64 + 32 * h + 8 * (i16 >> 2) + 4 * (wave & 1) + (i16 & 3)) * p.ldb * 2 + c * 16);
      }
    }
  }
  // CV: the two rows (g = 0, 1) of half-tile h this lane moves, for the tile whose first computed row is bml: byte offsets of their
  // padded-input pixels (+ the lane's 16-byte chunk, as above).  The padded input stays below 4 GB (host check).
  auto conv_rows = [&](int h, int bml) {
    if constexpr (CV) {
      const int HW = p.cv_H * p.cv_W;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int c = (lane & 7) ^ ((q + 4 * g) & 7);
        const int m = bml + wr * GS + h * A1_ROW + 16 * (wave & 3) + 8 * g + dr;
        const int t = m / HW, rem = m - t * HW;
        const int y = rem / p.cv_W, x = rem - y * p.cv_W;
        voA[h][g] = (unsigned)((((long long)t * (p.cv_H + 2) + y) * (p.cv_W + 2) + x) * p.cv_C * 4 + c * 16);
      }
    }
  };
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  unsigned char* dma_dst = lds + wave * 2048;                      // + buffer + half-tile + g * 1024
  constexpr int B_STEP = FA ? 64 : 128;                            // bytes of one K step in a B row (A: 128 in every mode)
  auto issue = [&](const char* base, const unsigned (&off)[2], int kbytes, int dst_off) {
    PP_GLDS(base + off[0] + kbytes, dma_dst + dst_off);
    PP_GLDS(base + off[1] + kbytes, dma_dst + dst_off + 1024);
  };
  // LNF: statistics of tile i (clamped to this workgroup's last tile) -> slot; wavefront w moves rows 32 w .. 32 w + 31 (64 floats)
  auto issue_stats = [&](int i, int slot) {
    const PPTile t = tile_entry(i < n_my ? i : n_my - 1);
    const float* src = p.ln_stats + ((long long)t.bml + 32 * wave) * 2 + lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + PP_STATS + slot * 2048 + wave * 256), 4, 0, 0);
  };
  // ---- fragment read addresses -----------------------------------------------------------------------------
  const unsigned rd0 = FA ? (unsigned)(l15 * 128 + (((2 * q) ^ sw) << 4))      // floats 8q .. 8q+3 of the row; 8q+4 .. 8q+7 = ^ 16
                          : (unsigned)(l15 * 128 + ((q ^ sw) << 4));            // k block 0; k block 1 = ^ 64
  const unsigned a_rd = (FH ? (unsigned)(l15 * 128 + ((q ^ sw) << 4)) : rd0) + wr * 64 * 128;   // FH: the converted image (convert_a) reads like an fp16 tile
  const unsigned b_rd = FA ? (unsigned)(wc * 2 * 2048 + l15 * 64 + q * 16)      // + e * 2048 + plane * 1024
                           : rd0 + wc * 32 * 128;                               // callers add the B0 slot (2 * PP_HT)

  f32x4 acc[8][4];
  f16x8 af[4][2], bf0[2][2], bf1[2][2];                            // FA: second index = bf16 plane (0 hi, 1 lo) instead of k block
  // FH (round 5): the f32 rows of an A half-tile are split into fp16 hi | lo ONCE, in place, by the wavefront whose DMA wrote them
  // (convert_a below) -- before that every wavefront split its whole 128 x 32 fragment itself, i.e. each element four times per tile (once
  // per wavefront column) on the vector pipe the partner wavefront's MFMAs issue on.  Same expressions, same values: bit-identical results.
  // Wavefront w owns LDS rows 16 w .. 16 w + 15 of every half-tile (its two DMA instructions); lane -> row lane >> 2, floats 8 j .. 8 j + 7
  // (j = lane & 3 = source chunks 2 j, 2 j + 1 at slots (2 j) ^ s, (2 j + 1) ^ s, s = (row >> 1) & 7).  The 8 hi halfs go to slot j ^ s, the
  // 8 lo halfs to slot j ^ s ^ 4: a row then reads [hi k 0..31 | lo k 0..31] under the fp16 kernel's swizzle -- k block 0 / 1 of read_a.
  // hi = fp16(s v), lo = fp16(s v - hi) (s v is exact, s v - hi too: one rounding each); the arithmetic is plain C between the two asm blocks
  // (round 4 lost a day to v_fma_mix in inline asm sunk next to MFMAs: profiles/r04/fp16x2_asm_hazard.txt); the reads / writes are asm so that
  // hipcc does not drain the DMA queue in front of them (see tile_entry).  Issuing the conversion's reads ahead of the phase's own fragment
  // reads and DMA issue (wait first, two fewer younger operations) was measured: no faster (profiles/r05/fp16x2_convert_once.txt).
  // In place is safe: a wavefront's reads complete (lgkmcnt) before its writes, and no other wavefront touches these rows between the
  // vmcnt wait that lands them and the barrier that publishes them (the barrier the DMA'd rows were published by before).
  auto convert_a = [&](unsigned buf, int h) {
    if constexpr (FH) {
      const int r = lane >> 2, j = lane & 3, s_ = (r >> 1) & 7;
      const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds) + buf + h * PP_HT + wave * 2048 + r * 128;
      const unsigned s0 = base + (((2 * j) ^ s_) << 4), s1 = base + (((2 * j + 1) ^ s_) << 4);
      const unsigned d0 = base + ((j ^ s_) << 4), d1 = base + ((j ^ s_ ^ 4) << 4);
      f32x4 x0, x1;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(s0), "v"(s1) : "memory");
      f16x8 h0, h1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? x0[e & 3] : x1[e & 3];
        const _Float16 a0 = (_Float16)(v * p.a_scale);
        h0[e] = a0; h1[e] = (_Float16)__builtin_fmaf(v, p.a_scale, -(float)a0);
      }
      asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" :: "v"(d0), "v"(d1), "v"(h0), "v"(h1) : "memory");
    }
  };
  auto read_a = [&](unsigned buf, int i) {
    if constexpr (FA && !FH) {
      using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if (i == 1 && mb < MB1_LO) continue;                         // TM = 192: only the upper two row blocks of half-tile 1
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(lds + buf + i * PP_HT + mb * 2048 + a_rd);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(lds + buf + i * PP_HT + mb * 2048 + (a_rd ^ 16));
        {
        bf16x8 h0, h1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = e < 4 ? x0[e & 3] : x1[e & 3];
          const __bf16 a0 = (__bf16)v;
          h0[e] = a0; h1[e] = (__bf16)(v - (float)a0);
        }
        af[mb][0] = __builtin_bit_cast(f16x8, h0); af[mb][1] = __builtin_bit_cast(f16x8, h1);
        }
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if (i == 1 && mb < MB1_LO) continue;
        af[mb][0] = *reinterpret_cast<const f16x8*>(lds + buf + i * PP_HT + mb * 2048 + a_rd);
        af[mb][1] = *reinterpret_cast<const f16x8*>(lds + buf + i * PP_HT + mb * 2048 + (a_rd ^ 64));
      }
    }
  };
  auto read_b = [&](unsigned buf, int j, f16x8 (&bf)[2][2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      bf[e][0] = *reinterpret_cast<const f16x8*>(lds + buf + j * PP_HT + e * 2048 + b_rd);
      bf[e][1] = *reinterpret_cast<const f16x8*>(lds + buf + j * PP_HT + e * 2048 + (FA ? b_rd + 1024 : (b_rd ^ 64)));
    }
  };
  auto mma = [&](int i, int j, const f16x8 (&bf)[2][2]) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (FA) {
      using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
      constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};        // a1 b0, a0 b1, a0 b0: smallest terms first
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (i == 1 && mb < MB1_LO) continue;
            const int ar = i * 4 + mb - (i == 1 ? MB1_LO : 0);       // accumulator row block
            if constexpr (FH)
              acc[ar][j * 2 + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[e][PB2[t]], af[mb][PA2[t]], acc[ar][j * 2 + e], 0, 0, 0);
            else
            acc[ar][j * 2 + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, bf[e][PB2[t]]), __builtin_bit_cast(bf16x8, af[mb][PA2[t]]), acc[ar][j * 2 + e], 0, 0, 0);
          }
    } else {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (i == 1 && mb < MB1_LO) {
          } else if constexpr (X3) {
            using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
            acc[i * 4 + mb][j * 2 + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, bf[e][kb]), __builtin_bit_cast(bf16x8, af[mb][kb]), acc[i * 4 + mb][j * 2 + e], 0, 0, 0);
          } else {
            const int ar = i * 4 + mb - (i == 1 ? MB1_LO : 0);
            acc[ar][j * 2 + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[e][kb], af[mb][kb], acc[ar][j * 2 + e], 0, 0, 0);
          }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the two DMA cursors: c1 = next K step of the half-tiles B1, A1; c2 = next K step of A0, B0 (one step further ahead) ----
  // (X3: a cursor also walks the six plane pairs; pair t reads plane PA[t] of A and PB[t] of B, 2 bits each in the codes below)
  struct Cur { int i, kt, k, t; const char *ta, *tb, *a, *b; int kin, tap; unsigned koff; };   // tile ordinal, K step of the tile, K step of the pair, pair
  // (CV: kin = K step inside the tap, tap = 3 kh + kw, koff = byte offset of this K step inside a padded-input pixel row walk)
  auto planes = [&](Cur& c) {
    if constexpr (X3) { c.a = c.ta + ((82 >> (2 * c.t)) & 3) * p.planeA; c.b = c.tb + ((280 >> (2 * c.t)) & 3) * p.planeB; }
    else { c.a = c.ta; c.b = c.tb; }
  };
  Cur c1, c2;
  { const PPTile t = tile_entry(0); c1.i = c1.kt = c1.k = c1.t = 0; c1.kin = c1.tap = 0; c1.koff = 0; c1.ta = CV ? Ab : Ab + t.a_off; c1.tb = Bb + t.b_off;
    planes(c1); c2 = c1; conv_rows(0, t.bml); conv_rows(1, t.bml); }
  // (tile change: called AFTER a phase's MFMAs were issued, where the wavefront has nothing else to do)
  auto advance = [&](Cur& c, int h) {                                 // h: the A half-tile this cursor feeds (c2: 0, c1: 1)
    ++c.k;
    if (X3 && c.k == nk1) { c.k = 0; ++c.t; if (c.t < 6) planes(c); }
    if constexpr (CV) {
      c.koff += 128;
      if (++c.kin == (p.cv_C >> 5)) { c.kin = 0; ++c.tap; c.koff = (unsigned)(((c.tap / 3) * (p.cv_W + 2) + c.tap % 3) * p.cv_C * 4); }
    }
    if (++c.kt == nk) {
      c.kt = c.k = c.t = 0;
      if constexpr (CV) { c.kin = c.tap = 0; c.koff = 0; }
      if (++c.i < n_my) {
        const PPTile t = tile_entry(c.i);
        c.ta = CV ? Ab : Ab + t.a_off; c.tb = Bb + t.b_off;
        conv_rows(h, t.bml);
      }
      planes(c);
    }
  };
  auto advance1 = [&]() { advance(c1, 1); };
  auto advance2 = [&]() { advance(c2, 0); };
  // (past the last tile the cursors keep re-loading the last tile's rows into slots nobody reads: the vmcnt counts stay uniform)
  auto issue_b1 = [&](unsigned buf) { issue(c1.b, voB[1], c1.k * B_STEP, buf + 3 * PP_HT); };
  auto issue_a1 = [&](unsigned buf) { issue(c1.a, voA[1], CV ? (int)c1.koff : c1.k * 128, buf + 1 * PP_HT); };
  auto issue_a0 = [&](unsigned buf) { issue(c2.a, voA[0], CV ? (int)c2.koff : c2.k * 128, buf + 0 * PP_HT); };
  auto issue_b0 = [&](unsigned buf) { issue(c2.b, voB[0], c2.k * B_STEP, buf + 2 * PP_HT); };

  if (p.desync_ns > 0) {
    // lab knob: start offsets (dbg bit 3 clear: the 32 workgroups of every XCD spread over [0, desync_ns); set: whole XCDs)
    unsigned long long wait = (unsigned long long)p.desync_ns * (((blockIdx.x >> 3) & 31u) * 8u + (blockIdx.x & 7u)) / 2560;   // 100 MHz ticks
    if (p.dbg & 8) wait = (unsigned long long)p.desync_ns * (blockIdx.x & 7u) / 80;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }

  // ---- epilogue of one tile: a lane owns one output row per 16-row block and 8 consecutive columns per column pair ----
  constexpr int ESZ = OUT == 0 ? 4 : 2;
  const unsigned lane_c = (EPI & 1) ? (unsigned)(((long long)(wr * GS + (l15 & 7)) * p.ldc + wc * 64 + 32 * (l15 >> 3) + 8 * q) * ESZ)
                                    : (unsigned)(((long long)(wr * GS + l15) * p.ldc + wc * 64 + 8 * q) * ESZ);
  // Row block outer, column pair inner: the two 64-byte halves of a 128-byte line come from CONSECUTIVE store instructions.
  // (Column pair outer -- the halves 8 instructions apart -- measured 5-6 % slower on the whole GEMM: partial-line writes.)
  auto epilogue_rows = [&](auto pred_tag, int bm, int bml, int bn, int bnl, int stats_slot) {
    constexpr bool PRED = decltype(pred_tag)::value;                 // edge tile: mask the rows / columns of the neighbour tile
    const int row0 = bml + wr * GS + l15, col0 = bnl + wc * 64 + 8 * q;
    // wave-uniform tile base + 32-bit lane offset (one VGPR live across the K loop; 256 rows of C stay far below 4 GB)
    char* cp = reinterpret_cast<char*>(p.C) + ((long long)bml * p.ldc + bnl) * ESZ;
    unsigned row_step = (unsigned)(16 * p.ldc * ESZ);
    unsigned lane_off = lane_c;
    const bool second = DUAL && bnl >= p.r_col0;                      // wave-uniform: this tile's columns belong to C2
    if constexpr (DUAL) {
      if (second) {
        cp = reinterpret_cast<char*>(p.C2) + ((long long)bml * p.ldc2 + (bnl - p.r_col0)) * ESZ;
        row_step = (unsigned)(16 * p.ldc2 * ESZ);
        lane_off = (unsigned)(((long long)(wr * GS + l15) * p.ldc2 + wc * 64 + 8 * q) * ESZ);
      }
    }
    asm volatile("" : "+v"(lane_off));                               // opaque: the per-row offsets are not worth 16 registers across the K loop
    // Every store below must be ISSUED by every wavefront, with at least one active lane: the counted vmcnt waits of the K loop
    // count them as younger operations.  Masked lanes of an edge tile therefore write to the library's dump buffer instead of
    // being switched off (a wavefront whose lanes are all masked would otherwise skip the instruction).
    char* dump = reinterpret_cast<char*>(p.dump) + lane * 32;
    if constexpr (FH) {
      // An activation beyond the fp16 range became hi = inf, lo = -inf: EVERY output of its row is then NaN, and a lane's accumulators of one
      // 16-row block all belong to one row -- one element per block tells (x * 0 != 0 for inf and NaN).  The accumulators go back to
      // scale 1 (exact) where they are consumed (put / the LayerNorm sums): scaling all 128 up front costs ~30 registers and spills.
      float chk = 0.f;
#pragma unroll
      for (int mb = 0; mb < MBT; ++mb) chk = __builtin_fmaf(acc[mb][0][0], 0.f, chk);
      if (chk != chk) {
        const unsigned faddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_FLAG);
        const int one = 1;
        asm volatile("ds_write_b32 %0, %1" :: "v"(faddr), "v"(one) : "memory");
      }
    }
    auto act4 = [&](f32x4& x) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (ACT == 1) x[e] = fmaxf(x[e], 0.f);
        else if constexpr (ACT == 2) x[e] = ovis::quick_gelu(x[e]);
        else if constexpr (ACT == 3) x[e] = ovis::gelu_erf(x[e]);
      }
    };
    auto pack8 = [&](const f32x4& x0, const f32x4& x1) {
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
      const f16x2 h0 = __builtin_convertvector(f32x2{x0[0], x0[1]}, f16x2), h1 = __builtin_convertvector(f32x2{x0[2], x0[3]}, f16x2);
      const f16x2 h2 = __builtin_convertvector(f32x2{x1[0], x1[1]}, f16x2), h3 = __builtin_convertvector(f32x2{x1[2], x1[3]}, f16x2);
      return make_uint4(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3));
    };
    auto store16 = [&](char* c, const uint4& o) {
      using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
      const u32x4 d = {o.x, o.y, o.z, o.w};
      if constexpr (((EPI >> 1) & 3) == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(c), "v"(d) : "memory");
      else if constexpr (((EPI >> 1) & 3) == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(c), "v"(d) : "memory");
      else *reinterpret_cast<uint4*>(c) = o;
    };
    // residual of the modes that add it in the epilogue (X3, FH): wave-uniform tile base + a 32-bit lane offset computed HERE (opaque to the
    // optimiser: hoisted out of the tile loop, the per-row-block 64-bit addresses are loop invariants that get spilled)
    const char* rb = nullptr;
    unsigned lane_r = 0, rstep = 0, r_row0 = 0;
    if constexpr (DUAL) {
      // rows repeat every r_mod rows (>= 256 = a tile's height: one wrap at most); a first-output tile loads too (from column 0, discarded):
      // every wavefront issues the same number of vector-memory operations per tile, which the counted waits of the K loop rely on
      rb = reinterpret_cast<const char*>(p.R) + (long long)(second ? bnl - p.r_col0 : 0) * 4;
      r_row0 = (unsigned)(bml % p.r_mod) + (unsigned)(wr * GS + l15);
      lane_r = (unsigned)((wc * 64 + 8 * q) * 4);
      rstep = (unsigned)(p.ldr * 4);
      asm volatile("" : "+v"(lane_r), "+v"(r_row0));
    } else if constexpr ((X3 || FH) && HAS_R) {
      rb = reinterpret_cast<const char*>(p.R) + ((long long)bml * p.ldr + bnl) * 4;
      lane_r = (unsigned)(((long long)(wr * GS + l15) * p.ldr + wc * 64 + 8 * q) * 4);
      rstep = (unsigned)(16 * p.ldr * 4);
      asm volatile("" : "+v"(lane_r));
    }
    auto rload = [&](int mb, int col) {                              // 4 residual values of row block mb at column col of the lane's 64
      if constexpr (DUAL) {
        unsigned row = r_row0 + (unsigned)(mb * 16);
        row = row >= (unsigned)p.r_mod ? row - (unsigned)p.r_mod : row;
        row = second ? row : (unsigned)l15;                            // (a first-output tile's discarded loads stay on 16 cached rows)
        const f32x4 v = *reinterpret_cast<const f32x4*>(rb + (row * rstep + lane_r + (unsigned)(col * 4)));   // R stays below 4 GB (host check)
        return second ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        return *reinterpret_cast<const f32x4*>(rb + (lane_r + (unsigned)mb * rstep + (unsigned)(col * 4)));
      }
    };
    // EPI bit 0: one 16-row block as two full-line stores.  low = lanes whose row is 0-7 of the block.  A: rows 0-7 (low lanes their own
    // columns 8q.., high lanes the low partner's columns 32 + 8q..), B: rows 8-15 (low lanes the high partner's columns 8q.., high lanes own).
    f32x2 lst[8];                                                    // LNF: (mean, rstd) of this lane's eight rows; c of its 16 columns
    f32x4 lc[4];
    if constexpr (LNF) {
      stats8(stats_slot, lst);
      bias8(bnl + wc * 64 + 8 * q, lc[0], lc[1]);
      bias8(bnl + wc * 64 + 8 * q + 32, lc[2], lc[3]);
    }
    auto put_lines = [&](int mb) {
      f32x4 x0 = acc[mb][0], x1 = acc[mb][1], y0 = acc[mb][2], y1 = acc[mb][3];
      if constexpr (LNF) {
        const float rs = lst[mb][1];
        x0 = x0 * rs + lc[0]; x1 = x1 * rs + lc[1]; y0 = y0 * rs + lc[2]; y1 = y1 * rs + lc[3];
      }
      act4(x0); act4(x1); act4(y0); act4(y1);
      const uint4 P0 = pack8(x0, x1), P1 = pack8(y0, y1);
      if constexpr (PSTAT) {                                         // this lane: row l15 of block mb, 16 of the wavefront's 64 columns
        using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
        const f16x2 one2 = {(_Float16)1.f, (_Float16)1.f};
        const unsigned pk[8] = {P0.x, P0.y, P0.z, P0.w, P1.x, P1.y, P1.z, P1.w};
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f16x2 v = __builtin_bit_cast(f16x2, pk[e]);
          sm = __builtin_amdgcn_fdot2(v, one2, sm, false);
          sq = __builtin_amdgcn_fdot2(v, v, sq, false);
        }
        // the row's other 48 columns sit in the lanes 16 and 32 away (q): v_permlane16_swap / v_permlane32_swap, own + partner
        auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
        sm = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
        r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sq = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
        auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
        sm = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
        r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sq = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
        // rows that two (shifted) tiles compute get the same value twice; one store instruction per row block from every wavefront
        float* pp = p.ln_part + (((long long)(bml + wr * GS + mb * 16 + l15)) * p.ln_pslots + (bnl >> 6) + wc) * 2;
        if (q == 0) *reinterpret_cast<float2*>(pp) = make_float2(sm, sq);
      }
      const bool low = l15 < 8;
      uint4 snd = low ? P1 : P0, rcv;
      rcv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd.x, 0x128, 0xf, 0xf, false);
      rcv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd.y, 0x128, 0xf, 0xf, false);
      rcv.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd.z, 0x128, 0xf, 0xf, false);
      rcv.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd.w, 0x128, 0xf, 0xf, false);
      const uint4 dA = low ? P0 : rcv, dB = low ? rcv : P1;
      // lane_off of this variant: row wr GS + (l15 & 7), column wc 64 + 32 (l15 >> 3) + 8 q
      char* cA = cp + (lane_off + (unsigned)mb * row_step);
      char* cB = cA + 8 * p.ldc * ESZ;
      if constexpr (PRED) {
        const int colL = bnl + wc * 64 + 32 * (l15 >> 3) + 8 * q, rowL = bml + wr * GS + (l15 & 7) + mb * 16;
        if (!(colL >= bn && rowL >= bm)) cA = dump;
        if (!(colL >= bn && rowL + 8 >= bm)) cB = dump;
      }
      store16(cA, dA);
      store16(cB, dB);
    };
    auto put = [&](int mb, int j, const f32x4* pre = nullptr) {      // pre: the two residual vectors of (mb, j), loaded by the caller
      f32x4 x0 = acc[mb][2 * j], x1 = acc[mb][2 * j + 1];
      if constexpr (FH && !LNO) { x0 *= p.out_scale; x1 *= p.out_scale; }
      if constexpr (X3 || (FH && !LNO)) {                            // bias / residual enter here, not as the accumulators' start value
        f32x4 b0, b1;
        bias8(col0 + 32 * j, b0, b1);
        x0 += b0; x1 += b1;
        if constexpr (HAS_R) {
          x0 += pre ? pre[0] : rload(mb, 32 * j);
          x1 += pre ? pre[1] : rload(mb, 32 * j + 4);
        }
      }
      act4(x0); act4(x1);
      char* c = cp + (lane_off + (unsigned)mb * row_step + (unsigned)(j * 32 * ESZ));
      if constexpr (PRED) { if (!(col0 + 32 * j >= bn && row0 + mb * 16 >= bm)) c = dump; }
      if constexpr (OUT == 1) {
        const uint4 o = pack8(x0, x1);
        store16(c, o);
      } else if constexpr (OUT == 2) {                               // exact 3-way bf16 split of the f32 result, one 16-byte store per plane
        using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
        bf16x8 h0, h1, h2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = e < 4 ? x0[e & 3] : x1[e & 3];
          const __bf16 a0 = (__bf16)v;
          const float r1 = v - (float)a0;
          const __bf16 a1 = (__bf16)r1;
          h0[e] = a0; h1[e] = a1; h2[e] = (__bf16)(r1 - (float)a1);
        }
        const long long pc = (PRED && c == dump) ? 0 : p.planeC;
        *reinterpret_cast<bf16x8*>(c) = h0;
        *reinterpret_cast<bf16x8*>(c + pc) = h1;
        *reinterpret_cast<bf16x8*>(c + 2 * pc) = h2;
      } else {
        *reinterpret_cast<f32x4*>(c) = x0; *reinterpret_cast<f32x4*>(c + 16) = x1;
      }
    };
    if constexpr (LNO) {
      // phase 1: this wavefront's share (64 columns) of every row's sum and sum of squares -> LDS [row][wc]
      const unsigned pbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_BIAS + 4096) + (unsigned)((wr * GS + l15) * 32 + wc * 8);
#pragma unroll
      for (int mb = 0; mb < MBT; ++mb) {
        float sm = 0.f, sq = 0.f;
        if constexpr (FH) {                                            // accumulators started at zero: back to scale 1, + bias + residual
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 b0, b1;
            bias8(col0 + 32 * j, b0, b1);
            acc[mb][2 * j] = acc[mb][2 * j] * p.out_scale + b0 + rload(mb, 32 * j);
            acc[mb][2 * j + 1] = acc[mb][2 * j + 1] * p.out_scale + b1 + rload(mb, 32 * j + 4);
          }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float v = acc[mb][nb][e]; sm += v; sq = __builtin_fmaf(v, v, sq); }
        auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
        sm = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
        r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sq = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
        auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
        sm = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
        r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sq = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
        const f32x2 pr = {sm, sq};
        const unsigned addr = pbase + (unsigned)mb * 512u;              // 16 rows x 32 bytes per row block
        if (q == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(pr) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
      // phase 2: totals of the lane's rows, normalise, store
      const unsigned gaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_BIAS) + (unsigned)((p.N + col0) * 4);
      const unsigned rbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + PP_BIAS + 4096) + (unsigned)((wr * GS + l15) * 32);
#pragma unroll
      for (int mb = 0; mb < MBT; ++mb) {
        f32x4 t0, t1;                                                 // (sum, sq) of wavefront columns 0, 1 | 2, 3
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1) : "v"(rbase + (unsigned)mb * 512u) : "memory");
        const float S1 = (t0[0] + t0[2]) + (t1[0] + t1[2]), S2 = (t0[1] + t0[3]) + (t1[1] + t1[3]);
        const float mean = S1 * (1.f / 256.f);
        const float var = fmaxf(S2 * (1.f / 256.f) - mean * mean, 0.f);
        const float rstd = 1.f / sqrtf(var + p.lno_eps);
        // gamma / beta of the lane's 16 columns, re-read per row block (eight LDS reads in flight, one wait): held across the whole
        // epilogue they cost 32 registers on top of the accumulators and spill
        f32x4 gm[4], bt[4];
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:128\n\tds_read_b128 %3, %8 offset:144\n\t"
                     "ds_read_b128 %4, %8 offset:1024\n\tds_read_b128 %5, %8 offset:1040\n\tds_read_b128 %6, %8 offset:1152\n\tds_read_b128 %7, %8 offset:1168\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(gm[0]), "=&v"(gm[1]), "=&v"(gm[2]), "=&v"(gm[3]), "=&v"(bt[0]), "=&v"(bt[1]), "=&v"(bt[2]), "=&v"(bt[3]) : "v"(gaddr) : "memory");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (acc[mb][nb] - mean) * rstd * gm[nb] + bt[nb];
#pragma unroll
        for (int j = 0; j < 2; ++j) put(mb, j);
      }
    } else if constexpr (DUAL) {
      // the residual loads of four row blocks go out together, ahead of those blocks' stores: a load that follows a store in the in-order
      // vmcnt queue waits for it, and the compiler's wait for any load is vmcnt(0) here (LDS-DMA outstanding) -- one drain per batch
      // instead of one per load (16 per tile, ~25 us of a 40 us tile when the loads sat between the stores)
#pragma unroll
      for (int hb = 0; hb < MBT; hb += 4) {
        f32x4 rr[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (hb + u < MBT) rr[u][c] = rload(hb + u, 32 * (c >> 1) + 4 * (c & 1));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (hb + u < MBT) put(hb + u, j, &rr[u][2 * j]);
      }
    } else {
#pragma unroll
    for (int mb = 0; mb < MBT; ++mb) {
      if constexpr (EPI & 1) put_lines(mb);
      else {
#pragma unroll
        for (int j = 0; j < 2; ++j) put(mb, j);
      }
    }
    }
  };
  // ... followed by the start value of the next tile's accumulators: bias + residual (gemm_epilogue.h) or zero
  auto acc_init = [&](int bml, int bnl, int stats_slot) {
    if constexpr (LNF) {                                             // accumulators start at -mean[m] s[n]
      f32x2 st[8];
      stats8(stats_slot, st);
      const int col0 = bnl + wc * 64 + 8 * q;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 sv[2];
        bias8(p.N + col0 + 32 * j, sv[0], sv[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int mb = 0; mb < MBT; ++mb) acc[mb][2 * j + e] = sv[e] * (-st[mb][0]);
      }
    } else if constexpr (X3 || FH) {
#pragma unroll
      for (int mb = 0; mb < MBT; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (R16 && (EPI & 1)) {
      // full-line residual loads (the mirror image of put_lines): A = rows 0-7 of a 16-row block, B = rows 8-15; the lanes l and l ^ 8
      // then exchange the pack that belongs to the partner's accumulator row
      using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
      const _Float16* rp = reinterpret_cast<const _Float16*>(p.R) + (long long)(bml + wr * GS + (l15 & 7)) * p.ldr + bnl + wc * 64 + 32 * (l15 >> 3) + 8 * q;
      const int col0 = bnl + wc * 64 + 8 * q;
      const bool low = l15 < 8;
      f32x4 bv[2][2];
      bias8(col0, bv[0][0], bv[0][1]);
      bias8(col0 + 32, bv[1][0], bv[1][1]);
#pragma unroll
      for (int mb = 0; mb < MBT; ++mb) {
        u32x4 LA, LB;
        if constexpr (((EPI >> 1) & 3) != 0) {
          LA = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + (long long)mb * 16 * p.ldr));
          LB = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rp + (long long)(mb * 16 + 8) * p.ldr));
        } else {
          LA = *reinterpret_cast<const u32x4*>(rp + (long long)mb * 16 * p.ldr);
          LB = *reinterpret_cast<const u32x4*>(rp + (long long)(mb * 16 + 8) * p.ldr);
        }
        const u32x4 snd = low ? LB : LA;
        u32x4 rcv;
#pragma unroll
        for (int e = 0; e < 4; ++e) rcv[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd[e], 0x128, 0xf, 0xf, false);
        const f16x8 r0 = __builtin_bit_cast(f16x8, low ? LA : rcv), r1 = __builtin_bit_cast(f16x8, low ? rcv : LB);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          acc[mb][e] = f32x4{(float)r0[4 * e], (float)r0[4 * e + 1], (float)r0[4 * e + 2], (float)r0[4 * e + 3]} + bv[0][e];
          acc[mb][2 + e] = f32x4{(float)r1[4 * e], (float)r1[4 * e + 1], (float)r1[4 * e + 2], (float)r1[4 * e + 3]} + bv[1][e];
        }
      }
    } else if constexpr (R16) {
      const _Float16* rp = reinterpret_cast<const _Float16*>(p.R) + (long long)(bml + wr * GS + l15) * p.ldr + bnl + wc * 64 + 8 * q;
      const int col0 = bnl + wc * 64 + 8 * q;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 bv[2];
        bias8(col0 + 32 * j, bv[0], bv[1]);
#pragma unroll
        for (int mb = 0; mb < MBT; ++mb) {
          const f16x8 r = *reinterpret_cast<const f16x8*>(rp + (long long)mb * 16 * p.ldr + 32 * j);
#pragma unroll
          for (int e = 0; e < 2; ++e)
            acc[mb][2 * j + e] = f32x4{(float)r[4 * e], (float)r[4 * e + 1], (float)r[4 * e + 2], (float)r[4 * e + 3]} + bv[e];
        }
      }
    } else if constexpr (HAS_R) {
      const float* rp = p.R + (long long)(bml + wr * GS + l15) * p.ldr + bnl + wc * 64 + 8 * q;
      const int col0 = bnl + wc * 64 + 8 * q;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 bv[2];
        bias8(col0 + 32 * j, bv[0], bv[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int mb = 0; mb < MBT; ++mb)
            acc[mb][2 * j + e] = *reinterpret_cast<const f32x4*>(rp + (long long)mb * 16 * p.ldr + 32 * j + 4 * e) + bv[e];
      }
    } else {                                                         // accumulators start at the bias: nothing to add in the epilogue
      const int col0 = bnl + wc * 64 + 8 * q;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 bv[2];
        bias8(col0 + 32 * j, bv[0], bv[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int mb = 0; mb < MBT; ++mb) acc[mb][2 * j + e] = bv[e];
      }
    }
  };

  // prologue: K steps 0 and 1 completely (both buffers).  Every tile starts in this state: the DMA of the first two K
  // steps of the NEXT tile is issued before the epilogue's stores (the in-order vmcnt queue then lets the stores drain
  // under two K steps of compute: no load that is waited for before the end of K step 1 is younger than a store).
  if constexpr (LNF) { issue_stats(0, 0); issue_stats(1, 1); }      // oldest in the queue: retired by the first counted wait below
  issue_a0(0); issue_b0(0); advance2(); issue_b1(0); issue_a1(0); advance1();
  issue_a0(PP_BUF); issue_b0(PP_BUF); advance2(); issue_b1(PP_BUF); issue_a1(PP_BUF); advance1();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                 // A0, B0 of K step 0
  convert_a(0, 0);
  if (wr == 1) PP_BARRIER();                                       // wavefronts 4-7 (G1) run one slot behind wavefronts 0-3 (G0)
  PP_BARRIER();
  // read segment of phase 1 of the first K step
  read_b(2 * PP_HT, 0, bf0);
  read_a(0, 0);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                 // B1 of K step 0 (5 younger half-tiles)
  PP_BARRIER();

  // (MBT row blocks: 2 fp16 / 4 f32 / 6 plane stores each, + 2 fp16 / 4 f32 residual loads; the f32 residual case is clamped to 63 - 10)
  constexpr int NS_ = OUT == 1 ? ((R16 ? 4 : 2) + (PSTAT ? 1 : 0)) * MBT : OUT == 2 ? 6 * MBT : (HAS_R ? 8 : 4) * MBT;
  constexpr int NS = NS_ > 53 ? 53 : NS_;   // vm ops of one epilogue (+ residual loads) per lane; 10 + NS <= 63
#define PP_WAIT(n_first, n_later) do { if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_first) : "memory"); \
                                       else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_later) : "memory"); } while (0)
#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
  // vm ops YOUNGER than the half-tile a wait is for (8 = four half-tiles), per K step of the tile; NS = the previous
  // epilogue's stores, which sit behind the early-issued B1/A1 of K step 1:
  //   kt = 0: (phase 1: in the tile transition below), 8 + NS, 8 + NS;   kt = 1: 8 + NS, 8 + NS, 8;   kt >= 2: 8
  // (the first tile of a workgroup has no stores in front of it: PP_WAIT's first argument)
  constexpr int NS1 = NS + (LNF ? 1 : 0);                            // + the statistics DMA issued right behind the epilogue
  static_assert(10 + NS1 <= 63, "vmcnt is a 6-bit field");
  unsigned s = 0;                                                    // global K step counter (buffer = s & 1)
  { const PPTile t = tile_entry(0); acc_init(t.bml, t.bnl, 0); }
  for (int it = 0; it < n_my; ++it) {
    unsigned long long* st = (p.stamps && it < 16 && (wave & 3) == 0 && lane == 0) ? p.stamps + ((blockIdx.x * 16 + it) * 2 + wr) * 4 : nullptr;
    if (st) { st[0] = __builtin_amdgcn_s_memrealtime(); st[3] = __builtin_amdgcn_s_memtime(); }   // [3]: shader clock, for the in-kernel clock
    unsigned long long* ks = (st && it >= 2 && it < 6) ? p.stamps + 256 * 16 * 2 * 4 + (blockIdx.x * 2 + wr) * 64 + (it - 2) * 16 : nullptr;
    for (int kt = 0; kt < nk; ++kt, ++s) {
      const unsigned cur = (s & 1) * PP_BUF, oth = PP_BUF - cur;
      // ---- phase 1: quadrant (0,0) ----
      if (kt) {
        read_b(cur + 2 * PP_HT, 0, bf0);
        read_a(cur, 0);
        issue_b1(oth);
        if (kt == 1) PP_WAIT(8, 8 + NS1); else PP_VMCNT(8);
        PP_BARRIER();
      }
      mma(0, 0, bf0);
      PP_BARRIER();
      // ---- phase 2: quadrant (0,1) ----
      read_b(cur + 2 * PP_HT, 1, bf1);
      if (kt) issue_a1(oth);
      if (kt < 2) PP_WAIT(8, 8 + NS1); else PP_VMCNT(8);
      convert_a(cur, 1);                                             // FH: A1 of this K step has landed (the wait above): own rows f32 -> fp16 hi | lo
      PP_BARRIER();
      mma(0, 1, bf1);
      if (kt) advance1();
      PP_BARRIER();
      // ---- phase 3: quadrant (1,1) ----
      read_a(cur, 1);
      issue_a0(cur);
      PP_BARRIER();
      mma(1, 1, bf1);
      PP_BARRIER();
      // ---- phase 4: quadrant (1,0), no LDS reads ----
      issue_b0(cur);
      if (kt == 0) PP_WAIT(8, 8 + NS1); else PP_VMCNT(8);
      convert_a(oth, 0);                                             // FH: A0 of the NEXT K step (of this or the next tile) has landed
      PP_BARRIER();
      mma(1, 0, bf0);
      advance2();
      PP_BARRIER();
      if (ks && kt < 16) ks[kt] = __builtin_amdgcn_s_memrealtime();
    }
    if (st) st[1] = __builtin_amdgcn_s_memrealtime();

    // ---- tile transition.  B1 / A1 of the next tile's K step 1 go out first (slots of buffer (s+1) & 1: last read 2 and 3
    // phases ago), i.e. BEFORE this wavefront's (remaining) stores.  G0 then waits one slot for G1's last MFMAs so that BOTH
    // groups run their epilogue in the same slot (a wavefront issues a 1-KB store every ~100 ns whatever the others do: two groups
    // storing one after the other cost twice the time); G1 reads its phase 1 under G0's first MFMAs of the next tile.
    const PPTile t = tile_entry(it);
    const int bm = t.bm, bn = t.bn, bml = t.bml, bnl = t.bnl;        // rows / columns this tile owns (stores) and computed (shifted edge tiles)
    const bool has_next = it + 1 < n_my;
    int nbml = bml, nbnl = bnl;
    if (has_next) { const PPTile tn_ = tile_entry(it + 1); nbml = tn_.bml; nbnl = tn_.bnl; }
    { const unsigned nb1 = (s & 1) * PP_BUF ^ PP_BUF; issue_b1(nb1); issue_a1(nb1); advance1(); }
    // slots:  G0: [last MFMAs] | (idle)      | epilogue, phase-1 reads | MFMAs of phase 1 ...
    //         G1: [phase-4 reads] | last MFMAs | epilogue               | phase-1 reads   | MFMAs ...
    if (wr == 0) PP_BARRIER();
    if (wr == 1 && !(p.dbg & 16)) __builtin_amdgcn_s_setprio(1);     // the younger wavefronts 4-7 lose every arbitration against 0-3 otherwise
    if (bm == bml && bn == bnl) epilogue_rows(std::false_type{}, bm, bml, bn, bnl, it & 1);
    else epilogue_rows(std::true_type{}, bm, bml, bn, bnl, it & 1);
    if (has_next) acc_init(nbml, nbnl, (it + 1) & 1);
    if (st) st[2] = __builtin_amdgcn_s_memrealtime();
    if (wr == 1) { __builtin_amdgcn_s_setprio(0); PP_BARRIER(); }
    {                                                                // read segment of the next tile's phase 1 (fragment registers are free again)
      const unsigned cur = (s & 1) * PP_BUF;
      read_b(cur + 2 * PP_HT, 0, bf0);
      read_a(cur, 0);
      PP_VMCNT(10 + NS);                                             // B1 of the next tile's K step 0: 5 younger half-tiles + the epilogue
    }
    PP_BARRIER();
    // LNF: every wavefront has read slot it & 1 (its epilogue) before the barrier above: refill it with the statistics of tile it + 2.
    // Issued by every wavefront for every tile (past the last one: a reload of the last tile's rows), so the counts stay uniform.
    if constexpr (LNF) issue_stats(it + 2, it & 1);
  }
  if (wr == 0) PP_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (FH) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && *reinterpret_cast<volatile int*>(lds + PP_FLAG) != 0 && p.range_flag) *p.range_flag = 1;
  }
}

// One 4 KB device buffer per device, allocated at the first launch there and kept for the life of the process.
void* pp_dump_buffer() {
  static void* buf[64] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!buf[dev] && hipMalloc(&buf[dev], 4096) != hipSuccess) buf[dev] = nullptr;
  return buf[dev];
}

// Tile height of a launch: 192-row tiles cost ~0.8 of a 256-row tile (3/4 of the MFMAs and stores, the same DMA volume) -- worth it
// when they save whole rounds of 256 workgroups.  g_pp_tm: 0 automatic, 256 / 192 forced (lab).
int g_pp_tm = 0;
int pp_pick_tm(int M, int tiles_n) {
  if (g_pp_tm == 256 || g_pp_tm == 192) return g_pp_tm;
  const double c256 = (double)ovis::cdiv((long long)ovis::cdiv(M, 256) * tiles_n, 256);
  const double c192 = 0.8 * (double)ovis::cdiv((long long)ovis::cdiv(M, 192) * tiles_n, 256);
  return c192 < 0.95 * c256 && (long long)ovis::cdiv(M, 192) * tiles_n <= 256ll * PP_MAX_TILES ? 192 : 256;
}

int g_f16_gemm_mode = 1;          // 1: ping-pong kernel for eligible problems, 0: gemm_f16_256_kernel (gemm_f16.hip)
int g_pp_grp = 6;                 // raster: at most this many N tiles per column group
int g_pp_desync_ns = -1;          // < 0: automatic
int g_pp_dbg = 0;
int g_pp_epi = 5;              // epilogue variant of the fp16-output launches (template parameter EPI): full-line nt stores, measured
                               // +7 % QKV, +6 % fc1, +5-9 % out-proj, +2 % fc2 against EPI 0 in one process (profiles/r03/lab_epi*.txt)
unsigned long long* g_pp_stamps = nullptr;

}  // namespace

namespace ovis {

bool gemm_f16_pp_eligible(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K, const float* bias,
                          const float* residual, long long ldr, int out_f16, bool act_is_none) {
  if (g_f16_gemm_mode != 1) return false;
  if (!((out_f16 && !residual) || (!out_f16 && act_is_none))) return false;   // instantiated combinations
  const long long blocks256 = (long long)cdiv(M, 256) * cdiv(N, 256);
  if (blocks256 > 256ll * PP_MAX_TILES) return false;
  if (blocks256 < 256 || K % 64 != 0 || K < 128 || N % 8 != 0 || M < 256 || N < 256) return false;
  if (256 * lda * 2 >= (1ll << 31) || 256 * ldb * 2 >= (1ll << 31)) return false;                      // 32-bit DMA row offsets inside a tile
  // the kernel fills the LDS bias region for every column of N (zeros when bias is NULL), so the limit holds with or without a bias
  if (N > PP_MAX_BIAS_N || (bias && (reinterpret_cast<uintptr_t>(bias) & 15))) return false;
  if (reinterpret_cast<uintptr_t>(C) & 15) return false;
  if (out_f16 ? (ldc % 8 != 0) : (ldc % 4 != 0)) return false;
  if (residual && ((ldr % 4 != 0) || (reinterpret_cast<uintptr_t>(residual) & 15))) return false;
  return true;   // (act, residual, out dtype) combinations without an instantiation are rejected by gemm_f16_pp_launch's caller check below
}

// fp16 residual + fp16 output (the tower's fp16 residual stream): same shape rules; R_f16 16-byte aligned, ldr % 8 == 0
bool gemm_f16_pp_res16_eligible(const void* C, const void* R16, long long lda, long long ldb, long long ldc, long long ldr, int M, int N, int K,
                                const float* bias) {
  if (!gemm_f16_pp_eligible(C, lda, ldb, ldc, M, N, K, bias, nullptr, 0, 1, true)) return false;
  return R16 && ldr % 8 == 0 && ldr >= N && (reinterpret_cast<uintptr_t>(R16) & 15) == 0;
}

int gemm_f16_pp_res16_launch(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                             const float* bias, const void* R16, long long ldr, hipStream_t s, float* part = nullptr) {
  PPArgs p;
  p.ln_part = part; p.ln_pslots = 4 * (int)cdiv(N, 256);
  p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(B); p.C = C; p.bias = bias;
  p.R = reinterpret_cast<const float*>(R16);
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = 0;
  p.tiles_n = (int)cdiv(N, 256);
  const int tm = pp_pick_tm(M, p.tiles_n);
  p.tiles_m = (int)cdiv(M, tm); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  p.desync_ns = g_pp_desync_ns >= 0 ? g_pp_desync_ns : 24000;        // residual GEMM: see gemm_f16_pp_launch
  p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  p.dump = pp_dump_buffer();
  if (!p.dump) return fail(OVIS_EINVAL, "gemm_nt_f16 (ping-pong): cannot allocate the 4 KB dump buffer");
  p.planeA = p.planeB = p.planeC = 0;
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
#define PP_R16(E_) do { if (tm == 192) hipLaunchKernelGGL((gemm_f16_pp_kernel<1, 0, true, false, false, true, 192, E_>), dim3(grid), dim3(512), 0, s, p); \
                        else hipLaunchKernelGGL((gemm_f16_pp_kernel<1, 0, true, false, false, true, 256, E_>), dim3(grid), dim3(512), 0, s, p); } while (0)
  if (part) {
    if (g_pp_epi != 5 || N % 256 != 0) return fail(OVIS_EINVAL, "gemm_nt_f16_res16_stats: needs the full-line epilogue and N %% 256 == 0");
    if (tm == 192) hipLaunchKernelGGL((gemm_f16_pp_kernel<1, 0, true, false, false, true, 192, 5, false, true>), dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemm_f16_pp_kernel<1, 0, true, false, false, true, 256, 5, false, true>), dim3(grid), dim3(512), 0, s, p);
    return check_launch("gemm_nt_f16 (ping-pong, fp16 residual, row statistics)");
  }
  switch (g_pp_epi) {
    case 0: PP_R16(0); break; case 1: PP_R16(1); break; case 4: PP_R16(4); break; case 5: PP_R16(5); break;
    default: return fail(OVIS_EINVAL, "gemm_nt_f16 (ping-pong, fp16 residual): bad epilogue variant %d", g_pp_epi);
  }
#undef PP_R16
  return check_launch("gemm_nt_f16 (ping-pong, fp16 residual)");
}

int gemm_f16_pp_launch(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                       const float* bias, const float* residual, long long ldr, int act, int out_f16, hipStream_t s) {
  PPArgs p;
  p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(B); p.C = C; p.bias = bias; p.R = residual;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = act;
  p.tiles_m = (int)cdiv(M, 256); p.tiles_n = (int)cdiv(N, 256); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  // residual GEMMs: every round of tiles ends in a 128 MB burst (residual loads + C stores); starting the 32 workgroups of
  // an XCD spread over 24 us de-phases the bursts (+4 % out-proj, +1.5 % fc2; -1 % on the fp16-output shapes, hence only here)
  p.desync_ns = g_pp_desync_ns >= 0 ? g_pp_desync_ns : (residual ? 24000 : 0);
  p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  p.dump = pp_dump_buffer();
  if (!p.dump) return fail(OVIS_EINVAL, "gemm_nt_f16 (ping-pong): cannot allocate the 4 KB dump buffer");
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;              // one persistent workgroup per CU (MI355X: 256 CUs)
  p.planeA = p.planeB = p.planeC = 0;
#define PP_LAUNCH(O, A_, R_) hipLaunchKernelGGL((gemm_f16_pp_kernel<O, A_, R_, false>), dim3(grid), dim3(512), 0, s, p)
#define PP_LAUNCH_E(A_, E_) hipLaunchKernelGGL((gemm_f16_pp_kernel<1, A_, false, false, false, false, 256, E_>), dim3(grid), dim3(512), 0, s, p)
  const int epi = g_pp_epi;
  if (out_f16 && !residual && epi != 0 && (act == 0 || act == 2)) {
#define PP_E(E_) case E_: if (act == 0) PP_LAUNCH_E(0, E_); else PP_LAUNCH_E(2, E_); break;
    switch (epi) { PP_E(1) PP_E(2) PP_E(3) PP_E(4) PP_E(5) default: return fail(OVIS_EINVAL, "gemm_nt_f16 (ping-pong): bad epilogue variant %d", epi); }
#undef PP_E
  } else if (out_f16 && !residual) {
    if (act == 0) PP_LAUNCH(1, 0, false); else if (act == 1) PP_LAUNCH(1, 1, false);
    else if (act == 2) PP_LAUNCH(1, 2, false); else PP_LAUNCH(1, 3, false);
  } else if (!out_f16 && act == 0) {
    if (residual) PP_LAUNCH(0, 0, true); else PP_LAUNCH(0, 0, false);
  } else {
    return fail(OVIS_EINVAL, "gemm_nt_f16 (ping-pong): no instantiation for out_f16=%d act=%d residual=%d", out_f16, act, residual != nullptr);
  }
#undef PP_LAUNCH
#undef PP_LAUNCH_E
  return check_launch("gemm_nt_f16 (ping-pong)");
}

// ---- LayerNorm folded into the GEMM (LNF): A = the raw fp16 rows, Wg = gamma-folded fp16 weight, c / s f32 [N], stats f32 [M][2] ----
bool gemm_f16_pp_ln_eligible(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K, const float* c, const float* s,
                             const float* stats, int act) {
  if (g_pp_epi != 5 || !(act == 0 || act == 2) || !c || !s || !stats) return false;
  if (!gemm_f16_pp_eligible(C, lda, ldb, ldc, M, N, K, c, nullptr, 0, 1, act == 0)) return false;
  if (2 * N > PP_MAX_BIAS_N || (reinterpret_cast<uintptr_t>(s) & 15) || (reinterpret_cast<uintptr_t>(stats) & 3)) return false;
  const long long tiles = (long long)cdiv(M, 256) * cdiv(N, 256);
  return cdiv(tiles, 256) <= PP_MAX_TILES_LNF && K >= 128;
}

int gemm_f16_pp_ln_launch(const void* A, long long lda, const void* Wg, long long ldb, void* C, long long ldc, int M, int N, int K,
                          const float* c, const float* s_rows, const float* stats, int act, hipStream_t s) {
  PPArgs p;
  p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(Wg); p.C = C; p.bias = c; p.R = nullptr;
  p.ln_s = s_rows; p.ln_stats = stats;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = 0; p.M = M; p.N = N; p.K = K; p.act = act;
  p.tiles_m = (int)cdiv(M, 256); p.tiles_n = (int)cdiv(N, 256); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  p.desync_ns = g_pp_desync_ns >= 0 ? g_pp_desync_ns : 0;
  p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  p.dump = pp_dump_buffer();
  if (!p.dump) return fail(OVIS_EINVAL, "gemm_nt_f16_ln (ping-pong): cannot allocate the 4 KB dump buffer");
  p.planeA = p.planeB = p.planeC = 0;
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
  if (act == 0) hipLaunchKernelGGL((gemm_f16_pp_kernel<1, 0, false, false, false, false, 256, 5, true>), dim3(grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((gemm_f16_pp_kernel<1, 2, false, false, false, false, 256, 5, true>), dim3(grid), dim3(512), 0, s, p);
  return check_launch("gemm_nt_f16_ln (ping-pong, LayerNorm folded)");
}

// ---- f32-A mode (bf16x2): A f32 [M,K], W as bf16 planes [>=2][N][ldb] (the first two of ovis_split_f32_to_bf16x3) ----
bool gemm_f32a_pp_eligible(const float* A, long long lda, const void* W3, long long ldb, long long plane, const float* C, long long ldc,
                           int M, int N, int K, const float* bias, const float* residual, long long ldr, int act, bool long_k = false) {
  if (act < 0 || act > 3 || (residual && act > 1)) return false;          // instantiated: none / ReLU (+ residual), QuickGELU, GELU
  const long long tiles_n = cdiv(N, 256), blocks256 = (long long)cdiv(M, 256) * tiles_n;
  // long_k (the 3x3 convolutions, K = 9 Cin): 72 K steps per tile amortise the tile transition and the schedule's edge over the tiled kernel is
  // 1.3-1.4x, so a launch of 0.8 rounds or of 1.1 still wins (5 x 92 x 160 x 256: 0.27 ms against 0.39)
  if (blocks256 < (long_k ? 200 : 256) || blocks256 > 256ll * PP_MAX_TILES || M < 256 || N < 256) return false;
  if (!long_k && blocks256 * 10 < 256 * cdiv(blocks256, 256) * 7) return false;       // last round < 40 % full on top of one round (e.g. 288 tiles): 2 WGs / CU of gemm_f32x3_kernel win
  if (tiles_n * 256 * 100 > (long long)N * 135) return false;           // > 35 % of the columns computed for nothing (N = 288: 78 %); at 33 % (N = 384, 576) the 1.5x faster loop still wins
  if (K % 32 != 0 || K < 64 || N % 8 != 0) return false;
  if (lda % 4 != 0 || ldb % 8 != 0 || plane % 8 != 0 || ldc % 4 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W3) | reinterpret_cast<uintptr_t>(C)) & 15) return false;
  if (256 * lda * 4 >= (1ll << 31) || plane * 2 + 256 * ldb * 2 >= (1ll << 31)) return false;           // 32-bit DMA offsets inside a tile
  // the kernel fills the LDS bias region for every column of N (zeros when bias is NULL), so the limit holds with or without a bias
  if (N > PP_MAX_BIAS_N || (bias && (reinterpret_cast<uintptr_t>(bias) & 15))) return false;
  if (residual && ((ldr % 4 != 0) || (reinterpret_cast<uintptr_t>(residual) & 15))) return false;
  return true;
}

// fh != nullptr: the fp16x2 arithmetic (template parameter FH): W3 = the two fp16 planes of w * fh->w_scale (ovis_split_f32_to_f16x2)
int gemm_f32a_pp_launch(const float* A, long long lda, const void* W3, long long ldb, long long plane, float* C, long long ldc, int M, int N,
                        int K, const float* bias, const float* residual, long long ldr, int act, hipStream_t s,
                        const float* ln_gamma = nullptr, const float* ln_beta = nullptr, float ln_eps = 0.f, const F16x2* fh = nullptr) {
  PPArgs p;
  p.lno_gamma = ln_gamma; p.lno_beta = ln_beta; p.lno_eps = ln_eps;
  p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(W3); p.C = C; p.bias = bias; p.R = residual;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = act;
  p.planeA = 0; p.planeB = plane * 2; p.planeC = 0;                  // bytes
  if (fh) {
    if (N > 4096) return fail(OVIS_EINVAL, "gemm_nt_f32 (ping-pong, fp16x2): N = %d > 4096", N);
    p.a_scale = fh->a_scale; p.acc_scale = fh->a_scale * fh->w_scale; p.out_scale = 1.f / p.acc_scale; p.range_flag = fh->flag;
  }
  p.tiles_n = (int)cdiv(N, 256);
  // LNO runs on 192-row tiles only: with 128 accumulator registers (256-row tiles) the LayerNorm epilogue spills ~80 registers per tile,
  // which costs what the fused LayerNorm saves (BriVIS, M = 695 520: no gain); 192-row tiles cost <= 9 % more GEMM time there and spill 9
  const int tm = ln_gamma ? ((long long)cdiv(M, 192) * p.tiles_n <= 256ll * PP_MAX_TILES ? 192 : 0) : pp_pick_tm(M, p.tiles_n);
  if (tm == 0) return fail(OVIS_EINVAL, "gemm_nt_f32_w3_ln: too many tiles (M=%d)", M);
  p.tiles_m = (int)cdiv(M, tm); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  p.desync_ns = g_pp_desync_ns > 0 ? g_pp_desync_ns : 0; p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  p.dump = pp_dump_buffer();
  if (!p.dump) return fail(OVIS_EINVAL, "gemm_nt_f32 (ping-pong, f32 A): cannot allocate the 4 KB dump buffer");
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
#define PP_K(A_, R_, TM_, LNO_, FH_) hipLaunchKernelGGL((gemm_f16_pp_kernel<0, A_, R_, false, true, false, TM_, 0, false, false, LNO_, false, FH_>), dim3(grid), dim3(512), 0, s, p)
#define PP_LAUNCH(A_, R_) do { if (fh) { if (tm == 192) PP_K(A_, R_, 192, false, true); else PP_K(A_, R_, 256, false, true); } \
                               else { if (tm == 192) PP_K(A_, R_, 192, false, false); else PP_K(A_, R_, 256, false, false); } } while (0)
  if (ln_gamma) {                                                   // LayerNorm of the output rows in the epilogue (LNO): N == 256, residual, no activation
    if (N != 256 || !residual || act != 0 || !ln_beta) return fail(OVIS_EINVAL, "gemm_nt_f32_w3_ln: needs N == 256, a residual and act == 0");
    if (fh) PP_K(0, true, 192, true, true); else PP_K(0, true, 192, true, false);
    return check_launch(fh ? "gemm_nt_f32 (ping-pong, f32 A, fp16x2, LayerNorm epilogue)" : "gemm_nt_f32 (ping-pong, f32 A, bf16x2, LayerNorm epilogue)");
  }
  if (residual) { if (act == 1) PP_LAUNCH(1, true); else PP_LAUNCH(0, true); }
  else if (act == 1) PP_LAUNCH(1, false);
  else if (act == 2) PP_LAUNCH(2, false);
  else if (act == 3) PP_LAUNCH(3, false);
  else PP_LAUNCH(0, false);
#undef PP_LAUNCH
#undef PP_K
  return check_launch(fh ? "gemm_nt_f32 (ping-pong, f32 A, fp16x2)" : "gemm_nt_f32 (ping-pong, f32 A, bf16x2)");
}

// ---- DUAL (fp16x2): C1 = A W[0:col0]^T + bias, C2 = A W[col0:N]^T + bias + R[m % r_rows] (template parameter DUAL) ----
bool gemm_f32a_pp_dual_eligible(const float* A, long long lda, const void* H2, long long ldb, long long plane, const float* C1, long long ldc1,
                                const float* C2, long long ldc2, int M, int N, int K, const float* bias, const float* R, long long ldr, int r_rows,
                                int col0) {
  if (!A || !H2 || !C1 || !C2 || !R || M < 256 || K % 32 != 0 || K < 64 || N % 8 != 0) return false;
  if (col0 <= 0 || col0 % 256 != 0 || N - col0 < 256 || r_rows < 256 || M % r_rows != 0) return false;   // no tile straddles col0 (the last one is
  const long long tiles_n = cdiv(N, 256), blocks = (long long)cdiv(r_rows, 192) * (M / r_rows) * tiles_n;  // shifted to N - 256); whole periods
  if (blocks > 256ll * PP_MAX_TILES || N > PP_MAX_BIAS_N || N > 4096) return false;
  if (lda % 4 != 0 || ldb % 8 != 0 || plane % 8 != 0 || ldc1 % 4 != 0 || ldc2 % 4 != 0 || ldr % 4 != 0) return false;
  if (ldc1 < col0 || ldc2 < N - col0 || ldr < N - col0 || (long long)r_rows * ldr * 4 >= (1ll << 32)) return false;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(H2) | reinterpret_cast<uintptr_t>(C1) | reinterpret_cast<uintptr_t>(C2) |
       reinterpret_cast<uintptr_t>(R)) & 15) return false;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return false;
  if (256 * lda * 4 >= (1ll << 31) || plane * 2 + 256 * ldb * 2 >= (1ll << 31) || 256 * ldc1 * 4 >= (1ll << 32) || 256 * ldc2 * 4 >= (1ll << 32)) return false;
  return true;
}

int gemm_f32a_pp_dual_launch(const float* A, long long lda, const void* H2, long long ldb, long long plane, float* C1, long long ldc1, float* C2,
                             long long ldc2, int M, int N, int K, const float* bias, const float* R, long long ldr, int r_rows, int col0,
                             hipStream_t s, const F16x2& fh) {
  PPArgs p;
  p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(H2); p.C = C1; p.bias = bias; p.R = R;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc1; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = 0;
  p.planeA = 0; p.planeB = plane * 2; p.planeC = 0;
  p.C2 = C2; p.ldc2 = ldc2; p.r_mod = r_rows; p.r_col0 = col0;
  p.a_scale = fh.a_scale; p.acc_scale = fh.a_scale * fh.w_scale; p.out_scale = 1.f / p.acc_scale; p.range_flag = fh.flag;
  p.tiles_n = (int)cdiv(N, 256);
  p.dual_T = M / r_rows;
  const int tm = pp_pick_tm(r_rows, p.dual_T * p.tiles_n);
  p.tiles_m = (int)cdiv(r_rows, tm) * p.dual_T; p.n_tiles = p.tiles_m * p.tiles_n;
  p.grp_w = p.tiles_n; p.grp_rem = 0;
  p.desync_ns = 0; p.dbg = g_pp_dbg; p.stamps = nullptr;
  p.dump = pp_dump_buffer();
  if (!p.dump) return fail(OVIS_EINVAL, "gemm_nt_f32_h2_dual: cannot allocate the 4 KB dump buffer");
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
  if (tm == 192) hipLaunchKernelGGL((gemm_f16_pp_kernel<0, 0, true, false, true, false, 192, 0, false, false, false, false, true, true>), dim3(grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((gemm_f16_pp_kernel<0, 0, true, false, true, false, 256, 0, false, false, false, false, true, true>), dim3(grid), dim3(512), 0, s, p);
  return check_launch("gemm_nt_f32_h2_dual (ping-pong, f32 A, fp16x2, two outputs)");
}

// ---- CV: 3x3 / stride 1 / pad 1 convolution on the f32-A schedule, input already zero-padded ([T][H+2][W+2][Cin] f32) ----
bool conv3x3_pp_eligible(const float* xpad, const void* W3, long long plane, const float* y, int T, int H, int W, int Cin, int Cout,
                         const float* bias, int act) {
  if (!(act == 0 || act == 1) || Cin % 32 != 0 || Cin < 32 || T < 1 || H < 1 || W < 1) return false;
  const long long M = (long long)T * H * W, K = 9ll * Cin;
  if (M >= (1ll << 31) || (long long)T * (H + 2) * (W + 2) * Cin * 4 + (2ll * (W + 2) + 2) * Cin * 4 + 128 >= (1ll << 32)) return false;   // 32-bit row bases
  return gemm_f32a_pp_eligible(xpad, K, W3, K, plane, y, Cout, (int)M, Cout, (int)K, bias, nullptr, 0, act, true);
}

int conv3x3_pp_launch(const float* xpad, const void* W3, long long plane, float* y, int T, int H, int W, int Cin, int Cout, const float* bias,
                      int act, hipStream_t s, const F16x2* fh = nullptr) {
  const int M = T * H * W, N = Cout, K = 9 * Cin;
  PPArgs p;
  p.A = reinterpret_cast<const _Float16*>(xpad); p.B = reinterpret_cast<const _Float16*>(W3); p.C = y; p.bias = bias; p.R = nullptr;
  p.lda = K; p.ldb = K; p.ldc = N; p.ldr = 0; p.M = M; p.N = N; p.K = K; p.act = act;
  p.planeA = 0; p.planeB = plane * 2; p.planeC = 0;
  p.cv_H = H; p.cv_W = W; p.cv_C = Cin;
  if (fh) {
    if (N > 4096) return fail(OVIS_EINVAL, "conv3x3 (ping-pong, fp16x2): Cout = %d > 4096", N);
    p.a_scale = fh->a_scale; p.acc_scale = fh->a_scale * fh->w_scale; p.out_scale = 1.f / p.acc_scale; p.range_flag = fh->flag;
  }
  p.tiles_n = (int)cdiv(N, 256);
  const int tm = pp_pick_tm(M, p.tiles_n);
  p.tiles_m = (int)cdiv(M, tm); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  p.desync_ns = g_pp_desync_ns > 0 ? g_pp_desync_ns : 0; p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  p.dump = pp_dump_buffer();
  if (!p.dump) return fail(OVIS_EINVAL, "conv3x3 (ping-pong, f32 A): cannot allocate the 4 KB dump buffer");
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
#define PP_K(A_, TM_, FH_) hipLaunchKernelGGL((gemm_f16_pp_kernel<0, A_, false, false, true, false, TM_, 0, false, false, false, true, FH_>), dim3(grid), dim3(512), 0, s, p)
#define PP_CV(A_) do { if (fh) { if (tm == 192) PP_K(A_, 192, true); else PP_K(A_, 256, true); } \
                       else { if (tm == 192) PP_K(A_, 192, false); else PP_K(A_, 256, false); } } while (0)
  if (act == 1) PP_CV(1); else PP_CV(0);
#undef PP_CV
#undef PP_K
  return check_launch(fh ? "conv3x3 (ping-pong, f32 A, fp16x2, padded input)" : "conv3x3 (ping-pong, f32 A, bf16x2, padded input)");
}

// x [n] f32 -> planes [3][n] bf16 with x == p0 + p1 + p2 exactly, 8 elements per thread (16-byte stores)
__global__ void __launch_bounds__(256)
x3_split8_kernel(const float4* __restrict__ x, uint4* __restrict__ p0, uint4* __restrict__ p1, uint4* __restrict__ p2, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = x[2 * i], b = x[2 * i + 1];
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  union { __bf16 h[8]; uint4 u; } o0, o1, o2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h0 = (__bf16)v[e];
    const float r1 = v[e] - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    o0.h[e] = h0; o1.h[e] = h1; o2.h[e] = (__bf16)(r1 - (float)h1);
  }
  p0[i] = o0.u; p1[i] = o1.u; p2[i] = o2.u;
}

}  // namespace ovis

extern "C" int ovis_gemm_x3pp_eligible(int M, int N, int K, int has_bias) {
  const long long blocks256 = (long long)ovis::cdiv(M, 256) * ovis::cdiv(N, 256);
  return blocks256 >= 256 && blocks256 <= 256ll * PP_MAX_TILES && K % 64 == 0 && K >= 64 && N % 8 == 0 && M >= 256 && N >= 256 &&
         N <= PP_MAX_BIAS_N && 256ll * K * 2 < (1ll << 31);   // the bias region is zero-filled for N columns when there is no bias
}

extern "C" int ovis_split_f32_to_bf16x3_v8(const float* x, void* planes, long long n, ovis_stream_t stream) {
  OVIS_REQUIRE(x && planes && n > 0 && n % 8 == 0, "split_f32_to_bf16x3_v8: n must be a positive multiple of 8");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(planes)) & 15) == 0, "split_f32_to_bf16x3_v8: 16-byte alignment");
  char* pl = reinterpret_cast<char*>(planes);
  hipLaunchKernelGGL(ovis::x3_split8_kernel, dim3(ovis::cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<uint4*>(pl), reinterpret_cast<uint4*>(pl + n * 2),
                     reinterpret_cast<uint4*>(pl + n * 4), n / 8);
  return ovis::check_launch("split_f32_to_bf16x3_v8");
}

extern "C" int ovis_gemm_nt_bf16x3_planes(const void* A3, long long lda, long long planeA, const void* W3, long long ldb, long long planeB,
                                          void* C, long long ldc, long long planeC, int M, int N, int K, const float* bias,
                                          const float* residual, long long ldr, int act, int out_planes, ovis_stream_t stream) {
  OVIS_REQUIRE(A3 && W3 && C, "gemm_nt_bf16x3_planes: null pointer");
  OVIS_REQUIRE(ovis_gemm_x3pp_eligible(M, N, K, bias != nullptr), "gemm_nt_bf16x3_planes: shape not eligible (M=%d N=%d K=%d)", M, N, K);
  OVIS_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % (out_planes ? 8 : 4) == 0, "gemm_nt_bf16x3_planes: bad leading dimensions");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(A3) | reinterpret_cast<uintptr_t>(W3) | reinterpret_cast<uintptr_t>(C)) & 15) == 0 &&
               (planeA * 2) % 16 == 0 && (planeB * 2) % 16 == 0 && (!out_planes || (planeC * 2) % 16 == 0), "gemm_nt_bf16x3_planes: 16-byte alignment");
  OVIS_REQUIRE((act == 0 || act == 1) && !(out_planes && residual), "gemm_nt_bf16x3_planes: act must be none / ReLU; plane output has no residual");
  OVIS_REQUIRE(!residual || (ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(residual) & 15) == 0), "gemm_nt_bf16x3_planes: residual alignment");
  OVIS_REQUIRE(!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0, "gemm_nt_bf16x3_planes: bias alignment");
  PPArgs p;
  p.A = reinterpret_cast<const _Float16*>(A3); p.B = reinterpret_cast<const _Float16*>(W3); p.C = C; p.bias = bias; p.R = residual;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.act = act;
  p.planeA = planeA * 2; p.planeB = planeB * 2; p.planeC = planeC * 2;         // bytes
  p.tiles_m = (int)ovis::cdiv(M, 256); p.tiles_n = (int)ovis::cdiv(N, 256); p.n_tiles = p.tiles_m * p.tiles_n;
  const int groups = (int)ovis::cdiv(p.tiles_n, g_pp_grp > 0 ? g_pp_grp : p.tiles_n);
  p.grp_w = p.tiles_n / groups; p.grp_rem = p.tiles_n % groups;
  p.desync_ns = g_pp_desync_ns > 0 ? g_pp_desync_ns : 0; p.dbg = g_pp_dbg; p.stamps = g_pp_stamps;
  p.dump = pp_dump_buffer();
  OVIS_REQUIRE(p.dump, "gemm_nt_bf16x3_planes: cannot allocate the 4 KB dump buffer");
  const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
  hipStream_t s = (hipStream_t)stream;
#define PP_LAUNCH(O, A_, R_) hipLaunchKernelGGL((gemm_f16_pp_kernel<O, A_, R_, true>), dim3(grid), dim3(512), 0, s, p)
  if (out_planes) { if (act == 1) PP_LAUNCH(2, 1, false); else PP_LAUNCH(2, 0, false); }
  else if (residual) { if (act == 1) PP_LAUNCH(0, 1, true); else PP_LAUNCH(0, 0, true); }
  else { if (act == 1) PP_LAUNCH(0, 1, false); else PP_LAUNCH(0, 0, false); }
#undef PP_LAUNCH
  return ovis::check_launch("gemm_nt_bf16x3_planes");
}

// C (fp16) = A B^T + bias + R (fp16): out-proj / c_proj of a CLIP block on the fp16 residual stream.  Only the shapes the ping-pong kernel
// takes (ovis_gemm_nt_f16_res16_eligible); smaller problems go through ovis_gemm_nt_f16 with an f32 residual (openvis_amd/ops.py).
extern "C" int ovis_gemm_nt_f16_res16_eligible(const void* C, const void* R16, long long lda, long long ldb, long long ldc, long long ldr,
                                               int M, int N, int K, const float* bias) {
  return ovis::gemm_f16_pp_res16_eligible(C, R16, lda, ldb, ldc, ldr, M, N, K, bias) ? 1 : 0;
}

extern "C" int ovis_gemm_nt_f16_res16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                                      const float* bias, const void* R16, long long ldr, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B && C && R16, "gemm_nt_f16_res16: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt_f16_res16: bad sizes");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_nt_f16_res16: alignment");
  OVIS_REQUIRE(ovis::gemm_f16_pp_res16_eligible(C, R16, lda, ldb, ldc, ldr, M, N, K, bias),
               "gemm_nt_f16_res16: shape not taken by the ping-pong kernel (M=%d N=%d K=%d): use ovis_gemm_nt_f16 with an f32 residual", M, N, K);
  return ovis::gemm_f16_pp_res16_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, R16, ldr, (hipStream_t)stream);
}

// ovis_gemm_nt_f16_res16 that also writes part [M][4 * N/256][2] = (sum, sum of squares) of every 64-column piece of the fp16 output rows
extern "C" int ovis_gemm_nt_f16_res16_stats(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, int M, int N, int K,
                                            const float* bias, const void* R16, long long ldr, float* part, ovis_stream_t stream) {
  OVIS_REQUIRE(A && B && C && R16 && part, "gemm_nt_f16_res16_stats: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N && N % 256 == 0, "gemm_nt_f16_res16_stats: bad sizes (N %% 256 == 0)");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
               (reinterpret_cast<uintptr_t>(part) & 7) == 0, "gemm_nt_f16_res16_stats: alignment");
  OVIS_REQUIRE(ovis::gemm_f16_pp_res16_eligible(C, R16, lda, ldb, ldc, ldr, M, N, K, bias),
               "gemm_nt_f16_res16_stats: shape not taken by the ping-pong kernel (M=%d N=%d K=%d)", M, N, K);
  return ovis::gemm_f16_pp_res16_launch(A, lda, B, ldb, C, ldc, M, N, K, bias, R16, ldr, (hipStream_t)stream, part);
}

// C (fp16) = act( LayerNorm(A) W^T + b ) from the raw fp16 rows A (LayerNorm folded into the GEMM; include/openvis_hip.h).
extern "C" int ovis_gemm_nt_f16_ln_eligible(const void* C, long long lda, long long ldb, long long ldc, int M, int N, int K, const float* c,
                                            const float* s_rows, const float* stats, int act) {
  return ovis::gemm_f16_pp_ln_eligible(C, lda, ldb, ldc, M, N, K, c, s_rows, stats, act) ? 1 : 0;
}

extern "C" int ovis_gemm_nt_f16_ln(const void* A, long long lda, const void* Wg, long long ldb, void* C, long long ldc, int M, int N, int K,
                                   const float* c, const float* s_rows, const float* stats, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(A && Wg && C && c && s_rows && stats, "gemm_nt_f16_ln: null pointer");
  OVIS_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt_f16_ln: bad sizes");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Wg)) & 15) == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_nt_f16_ln: alignment");
  OVIS_REQUIRE(ovis::gemm_f16_pp_ln_eligible(C, lda, ldb, ldc, M, N, K, c, s_rows, stats, act),
               "gemm_nt_f16_ln: shape not taken by the ping-pong kernel (M=%d N=%d K=%d act=%d): use ovis_layernorm_f16_to_f16 + ovis_gemm_nt_f16", M, N, K, act);
  return ovis::gemm_f16_pp_ln_launch(A, lda, Wg, ldb, C, ldc, M, N, K, c, s_rows, stats, act, (hipStream_t)stream);
}

extern "C" int ovis_set_f16_gemm_mode(int mode, int raster_group, int desync_ns) {
  OVIS_REQUIRE(mode == 0 || mode == 1, "set_f16_gemm_mode: mode must be 0 (gemm_f16_256_kernel) or 1 (ping-pong kernel)");
  OVIS_REQUIRE(raster_group >= 0 && raster_group <= 64, "set_f16_gemm_mode: bad tuning value");
  g_f16_gemm_mode = mode;
  if (raster_group > 0) g_pp_grp = raster_group;
  g_pp_desync_ns = desync_ns;
  return OVIS_OK;
}

// lab only (tools/gemm_lab.cpp; not part of include/openvis_hip.h): debug flags and the in-kernel time stamp buffer
extern "C" int ovis_pp_debug(int flags, unsigned long long* stamps) { g_pp_dbg = flags; g_pp_stamps = stamps; return OVIS_OK; }
extern "C" int ovis_pp_epilogue(int epi) { g_pp_epi = epi; return OVIS_OK; }   // lab / tests: EPI variant 0-5
extern "C" int ovis_pp_tile_rows(int tm) { g_pp_tm = tm; return OVIS_OK; }   // lab / tests: 0 automatic, 256, 192
