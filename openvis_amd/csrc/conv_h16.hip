// fp16-activation convolutions of the ResNet backbone (A2) under the reference's autocast policy: 3x3 (stride 1 / 2, pad 1) and 1x1
// (stride 1 / 2) implicit GEMMs on fp16 NHWC activations with fp16 weights, f32 accumulation, bias (+ f32 residual) + ReLU epilogue,
// fp16 or f32 output.
//
//   y[t,oy,ox,co] = act( sum_{kh,kw,c} x[t, oy s - p + kh, ox s - p + kw, c] * w[co,kh,kw,c] + bias[co] + R[t,oy,ox,co] )
//
// Replaces the cuDNN calls behind detectron2's BottleneckBlock (configs/openvoc_ytvis/Base.yaml:2-16: ResNet-50, STRIDE_IN_1X1 False)
// as torch.cuda.amp.autocast runs them (train_net.py:241: fp16 operands, f32 accumulation).  Why a kernel of its own: the 3x3
// convolutions of the four stages are all ~21.7 GFLOP (M N K = const) and took 80-104 us each on the register-staged 64x64 kernel
// (gemm_f16cvt.hip: ~240 TFLOP/s, bound by the latency of its global -> register -> LDS staging with two K tiles in flight;
// profiles/r04/backbone_launches_before.txt) -- 1.45 ms of the 4.3 ms backbone.  Here:
//   * activations between the convolutions of a bottleneck are STORED in fp16 (conv1 -> conv2 -> conv3): the next convolution rounded
//     them to fp16 while staging anyway, so the values that reach the MFMA are bit-identical, the tensor is half the bytes and the
//     operand tile can go global -> LDS by LDS-DMA (global_load_lds_dwordx4) with no conversion pass;
//   * 128 x BN x 64 tiles (BN = 128, or 64 for Cout = 64), 4 wavefronts (2 x 2), v_mfma_f32_32x32x16_f16; a 3-slot LDS ring: the DMA of
//     K step s + 2 is issued right behind the barrier of step s, the only waits are counted (`s_waitcnt vmcnt(PER)`: one K step stays
//     in flight across every barrier), raw s_barrier -- one barrier per K step;
//   * implicit im2col without padding the input: a lane owns 4 of the tile's 128 pixels; per pixel one 32-bit byte offset of its top-left
//     tap and a 9-bit mask of the taps that fall inside the image, computed once; a K step (64 channels of one tap) adds a wave-uniform
//     offset, taps outside the image read a page of zeros instead (address select, no branch);
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 on the DMA's per-lane SOURCE address and on the
//     fragment reads (conflict-free ds_read_b128 for the 32-row fragments, as in gemm_f16_pp.hip);
//   * operand roles swapped (weights as the MFMA's A operand) so that a lane owns one output pixel and 4 x 4 consecutive channels:
//     the vectorised epilogue of gemm_epilogue.h (fp16: 16-byte stores via v_permlane32_swap).
#include "common.h"
#include "gemm_epilogue.h"
#include <mutex>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct CH16Args {
  const _Float16* X; const _Float16* Wt; void* Y; const float* bias; const float* R; const char* zeros;
  int T, H, W, Cin, OH, OW, Cout, stride, act, M, tiles_n;
};

#define CH_GLDS(src, dst) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// DBG (lab instantiations only, ovis_conv_h16_debug): 1 = no DMA after the prologue, 2 = no fragment reads / MFMAs.  A compile-time switch: as a
// run-time branch around compute() it made the register allocator carry the accumulators in VGPRs and copy all 64 to AGPRs and back per K step.
template <int BN, int TAPS, bool OUT16, int NST = 3, int DBG = 0>
__global__ void __launch_bounds__(256)
conv_h16_kernel(const CH16Args p) {
  constexpr int BM = 128;
  constexpr int STAGE = (BM + BN) * 128;                 // bytes of one K step: A rows, then B rows, 128 B each
  constexpr int BJ = BN / 32;                            // B DMA instructions per wavefront and K step (A: 4)
  constexpr int PER = 4 + BJ;
  constexpr int TM = 2, TN = BN / 64;                    // 32x32 accumulator tiles of a wavefront (64 x BN/2 outputs)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];   // ONE LDS object (a second one de-pipelines the DMA)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned bid = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const int bn = (int)(bid % p.tiles_n) * BN;
  const int bm = (int)(bid / p.tiles_n) * BM;
  const int K = TAPS * p.Cin;
  const int cpt = p.Cin >> 6;                            // K steps per tap
  const int nk = TAPS * cpt;

  // ---- this lane's four pixels (rows 32 wave + 8 j + lane / 8 of the tile): byte offset of tap (0, 0), mask of the taps inside the image
  int base[4];
  unsigned mask[4];
  const int dr = lane >> 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = min(bm + (4 * wave + j) * 8 + dr, p.M - 1);
    const int ox = m % p.OW, t1 = m / p.OW, oy = t1 % p.OH, t = t1 / p.OH;
    const int iy0 = oy * p.stride - (TAPS == 9 ? 1 : 0), ix0 = ox * p.stride - (TAPS == 9 ? 1 : 0);
    base[j] = ((t * p.H + iy0) * p.W + ix0) * p.Cin * 2;            // may be negative at the border: those taps are masked
    unsigned mk = 0;
    if constexpr (TAPS == 9) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if (iy0 + kh >= 0 && iy0 + kh < p.H && ix0 + kw >= 0 && ix0 + kw < p.W) mk |= 1u << (kh * 3 + kw);
    } else {
      mk = 1u;
    }
    mask[j] = mk;
  }
  // logical 16-byte chunk this lane fetches for LDS slot (row, lane & 7): (lane & 7) ^ ((row >> 1) & 7); row = 8 g + dr
  int cA[4], cB[BJ];
  long long wB[BJ];
#pragma unroll
  for (int j = 0; j < 4; ++j) cA[j] = ((lane & 7) ^ ((((4 * wave + j) * 8 + dr) >> 1) & 7)) * 16;
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = (wave * BJ + j) * 8 + dr;
    cB[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
    wB[j] = (long long)min(bn + row, p.Cout - 1) * K * 2 + cB[j];
  }
  // per-lane source pointers, complete except for the wave-uniform part of a K step (no 64-bit arithmetic beyond one add per DMA instruction,
  // no integer division in the loop: the issue cursor walks (tap, channel chunk) by increments)
  const char* pa[4];
  const char* pb[BJ];
#pragma unroll
  for (int j = 0; j < 4; ++j) pa[j] = reinterpret_cast<const char*>(p.X) + (long long)base[j] + cA[j];
#pragma unroll
  for (int j = 0; j < BJ; ++j) pb[j] = reinterpret_cast<const char*>(p.Wt) + wB[j];
  const char* Zb = p.zeros + (lane & 7) * 16;
  int i_tap = 0, i_kc = 0, i_kh = 0, i_kw = 0;        // the NEXT K step to issue: tap (kh, kw), channel chunk kc; all wave-uniform
  long long i_woff = 0;                              // its byte offset in a weight row

  auto issue = [&](int buf) {
    long long off;
    if constexpr (TAPS == 9) off = ((long long)(i_kh * p.W + i_kw) * p.Cin + i_kc * 64) * 2;
    else off = (long long)i_kc * 128;
    const unsigned tapbit = 1u << i_tap;
    unsigned char* dst = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* src = (mask[j] & tapbit) ? pa[j] + off : Zb;
      CH_GLDS(src, dst + (4 * wave + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) CH_GLDS(pb[j] + i_woff, dst + BM * 128 + (wave * BJ + j) * 1024);
    i_woff += 128;
    if (++i_kc == cpt) {
      i_kc = 0; ++i_tap;
      if (++i_kw == 3) { i_kw = 0; ++i_kh; }
    }
  };

  const int r32 = lane & 31, h = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets inside a stage: row * 128 + ((4 h + s) ^ ((row >> 1) & 7)) * 16
  unsigned aoff[TM], boff[TN], asw[TM], bsw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) { const int row = wr * 64 + i * 32 + r32; aoff[i] = row * 128; asw[i] = (row >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < TN; ++j) { const int row = wc * (BN / 2) + j * 32 + r32; boff[j] = BM * 128 + row * 128; bsw[j] = (row >> 1) & 7; }

  // fragment reads run ONE k chunk ahead of the MFMAs (two register sets): while the four / two MFMAs of chunk s run, the reads of chunk
  // s + 1 are in flight and the compiler's counted lgkmcnt wait retires only chunk s.  The order is pinned (sched_barrier): left alone, the
  // scheduler sinks every read group next to its MFMAs behind a full lgkmcnt(0) -- one exposed LDS round trip per chunk, four per K step.
  auto compute = [&](int buf) {
    const unsigned char* st = lds + buf * STAGE;
    f16x8 af[2][TM], bf[2][TN];
    auto rd = [&](int s, int set) {
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const f16x8*>(st + boff[j] + (((4 * h + s) ^ bsw[j]) << 4));
#pragma unroll
      for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const f16x8*>(st + aoff[i] + (((4 * h + s) ^ asw[i]) << 4));
    };
    rd(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) rd(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);   // roles swapped
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // NST slots: NST - 1 K steps in flight.  K step ks has landed for THIS wavefront once at most the (NST - 2) PER instructions of the later
  // steps are outstanding; the barrier then makes every wavefront's share visible -- and says that all of them are done reading slot
  // (ks - 1) % NST, which is refilled next (step ks + NST - 1).
  static_assert(NST == 2 || NST == 3, "2 or 3 LDS slots");
#pragma unroll
  for (int q = 0; q < NST - 1; ++q)
    if (q < nk) issue(q);
  for (int ks = 0; ks < nk; ++ks) {
    if (NST == 3 && ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if ((DBG & 1) == 0 && ks + NST - 1 < nk) issue((ks + NST - 1) % NST);
    if constexpr ((DBG & 2) == 0) compute(ks % NST);
  }

  const bool vec_ok = ovis::epilogue_vec_ok(p.Y, p.Cout, p.bias, p.R, p.Cout);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm + wr * 64 + i * 32 + r32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      ovis::epilogue_tile<OUT16>(acc[i][j], m, m < p.M, bn + wc * (BN / 2) + j * 32, h, p.Cout, p.Y, p.Cout, p.bias, p.R, p.Cout, p.act, vec_ok);
  }
}

int g_ch_nst = 0;         // LDS slots of conv_h16_kernel (lab switch ovis_conv_h16_slots): 0 = automatic; 3 = two K steps in flight (96 KB at BN = 128:
                         // one workgroup per CU); 2 = one in flight, two workgroups per CU
int g_ch_bn = 0;          // lab switch: 0 automatic, 64 / 128 forced
int g_ch_dbg = 0;

// One 4 KB page of zeros per device (what a masked tap reads), allocated at the first launch there and kept for the life of the process.
const char* zero_page() {
  static void* buf[64] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!buf[dev]) {
    if (hipMalloc(&buf[dev], 4096) != hipSuccess) { buf[dev] = nullptr; return nullptr; }
    if (hipMemset(buf[dev], 0, 4096) != hipSuccess) { (void)hipFree(buf[dev]); buf[dev] = nullptr; return nullptr; }
  }
  return reinterpret_cast<const char*>(buf[dev]);
}

// 2x2 / 3x3 stride-2 pad-1 max pool on fp16 NHWC (the stem's pool; max commutes with the fp16 rounding, so pooling the fp16 map gives
// exactly the fp16 rounding of the pooled f32 map)
__global__ void __launch_bounds__(256)
maxpool3x3s2_h16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, int OH, int OW) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * OH * OW * C8;
  if (i >= total) return;
  const int c = (int)(i % C8);
  long long r = i / C8;
  const int ow = (int)(r % OW); r /= OW;
  const int oh = (int)(r % OH);
  const int n = (int)(r / OH);
  union U { uint4 u; _Float16 h[8]; } best;
#pragma unroll
  for (int e = 0; e < 8; ++e) best.h[e] = (_Float16)(-65504.f);
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    if (ih < 0 || ih >= H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      if (iw < 0 || iw >= W) continue;
      U v;
      v.u = x[(((long long)n * H + ih) * W + iw) * C8 + c];
#pragma unroll
      for (int e = 0; e < 8; ++e) best.h[e] = v.h[e] > best.h[e] ? v.h[e] : best.h[e];
    }
  }
  y[i] = best.u;
}


// ---- ResNet stem + max pool in one kernel (round 4) ---------------------------------------------------------------------------------------
// conv 7x7 / stride 2 / pad 3 on the normalised NHWC4 frames (3 channels + a zero one; the kernel is padded to 7 x 8 taps: K = 7 * 8 * 4 = 224),
// FrozenBN folded, ReLU, rounded to fp16, then the 3x3 / stride 2 / pad 1 max pool (detectron2 BasicStem, configs/openvoc_ytvis/Base.yaml:2-16).
// As an implicit GEMM (gemm_f16cvt_kernel<64,64,ConvA>) every input pixel is fetched ~12 times through the load path (0.22 ms at 720p, all
// L2 -> L1 traffic) and the 150 MB conv map makes a round trip to memory for the pool.  Here a workgroup owns 4 x 16 POOLED pixels: it stages
// the 23 x 72 input pixels under them once (fp16, 13 KB of LDS), computes the 9 x 33 conv pixels the pool windows need on the MFMA -- the
// tile's 297 pixels flattened into 19 blocks of 16, weights as the A operand (64 channels x (kh: 8 taps x 4 channels = one K step of 32)) held
// in registers, pixels as the B operand read from the patch (a lane's 8 k = two neighbouring input pixels = 16 contiguous bytes) -- writes
// relu(acc + bias) as fp16 into a 38 KB LDS tile and pools from there.  Positions outside the conv map are written as 0: every pool window
// holds a valid position and all values are >= 0 after the ReLU, so 0 behaves as the pool's -inf padding.
constexpr int ST_PH = 4, ST_PW = 16;                   // pooled pixels per tile
int g_stem_xt = 2;                                     // tiles per workgroup (lab: ovis_stem_tiles)
constexpr int ST_CH = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1;        // conv pixels: 9 x 33
constexpr int ST_IH = 2 * ST_CH + 5, ST_IW = 2 * ST_CW + 6;        // input pixels: 23 x 72 (8-tap rows)
constexpr int ST_M = ST_CH * ST_CW, ST_MT = (ST_M + 15) / 16;      // 297 conv pixels, 19 blocks of 16

__global__ void __launch_bounds__(256)
stem_pool_kernel(const float4* __restrict__ x, const uint4* __restrict__ w16, const float* __restrict__ bias, uint4* __restrict__ y, int T, int H,
                 int W, int OH, int OW, int PH, int PW, int tiles_x, int tiles_y, int xt) {
  using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
  using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  __shared__ __attribute__((aligned(16))) f16x4 patch[ST_IH * ST_IW];          // [row][col] x 4 channels
  __shared__ __attribute__((aligned(16))) _Float16 conv[ST_M * 64];            // [conv pixel][channel]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  // a workgroup walks xt neighbouring tiles of one tile row with the 28 KB of weight fragments per wavefront fetched once.  Measured at
  // 5 x 736 x 1280 (tools/bench_stem.py, us): xt = 1: 131, 2: 126, 4: 140, 8: 170, 20: 194 -- the weight traffic (515 MB per clip at xt = 1)
  // is not what bounds the kernel, the number of workgroups in flight is; conv + pool as two launches: 228
  const int groups_x = (tiles_x + xt - 1) / xt;
  const int gx = blockIdx.x % groups_x, by = (blockIdx.x / groups_x) % tiles_y, t = blockIdx.x / (groups_x * tiles_y);

  // weights: A fragments of the 4 channel blocks x 7 kernel rows, 8 halfs each: W[n][kh][2 kq .. 2 kq + 1][0..3]
  uint4 wf[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[kh][nt] = w16[((nt * 16 + l15) * 7 + kh) * 4 + kq];

  const float4* xp = x + (long long)t * H * W;
  for (int bx = gx * xt; bx < min(tiles_x, (gx + 1) * xt); ++bx) {
  const int py0 = by * ST_PH, px0 = bx * ST_PW;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;                      // first conv pixel of the tile
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;                      // first input pixel of the patch
  // input patch -> LDS as fp16 (zero outside the frame: the convolution's padding)
  for (int idx = tid; idx < ST_IH * ST_IW; idx += 256) {
    const int r = idx / ST_IW, c = idx - r * ST_IW;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float4 v = xp[in ? (long long)iy * W + ix : 0];
    patch[idx] = in ? f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w} : f16x4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
  }
  __syncthreads();

  const float4 bz = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int mt = wave; mt < ST_MT; mt += 4) {
    const int m = mt * 16 + l15;                                       // this lane's conv pixel as the B operand's column
    const int mc = min(m, ST_M - 1);
    const int cyl = mc / ST_CW, cxl = mc - cyl * ST_CW;
    const f16x4* pb = patch + (2 * cyl) * ST_IW + 2 * cxl + 2 * kq;     // + kh * ST_IW: input row 2 cyl + kh, columns 2 cxl + 2 kq, + 1
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      const f16x8 xb = *reinterpret_cast<const f16x8*>(pb + kh * ST_IW);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[kh][nt]), xb, acc[nt], 0, 0, 0);
    }
    // D[n = 16 nt + 4 kq + j][m = l15]: four consecutive channels of one conv pixel per lane and channel block
    const bool ok = m < ST_M;
    const int cy = cy0 + cyl, cx = cx0 + cxl;
    const bool valid = cy >= 0 && cy < OH && cx >= 0 && cx < OW;
    if (ok) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 b4 = valid ? *reinterpret_cast<const float4*>(bias + nt * 16 + 4 * kq) : bz;
        f16x4 o;
        o[0] = (_Float16)(valid ? fmaxf(acc[nt][0] + b4.x, 0.f) : 0.f); o[1] = (_Float16)(valid ? fmaxf(acc[nt][1] + b4.y, 0.f) : 0.f);
        o[2] = (_Float16)(valid ? fmaxf(acc[nt][2] + b4.z, 0.f) : 0.f); o[3] = (_Float16)(valid ? fmaxf(acc[nt][3] + b4.w, 0.f) : 0.f);
        *reinterpret_cast<f16x4*>(&conv[m * 64 + nt * 16 + 4 * kq]) = o;
      }
    }
  }
  __syncthreads();

  // pool: thread -> (pooled pixel tid / 4, 16 channels)
  const int pp = tid >> 2, cg = (tid & 3) * 16;
  const int ppy = pp / ST_PW, ppx = pp - ppy * ST_PW;
  const int py = py0 + ppy, px = px0 + ppx;
  if (py < PH && px < PW) {
    f16x8 m0, m1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { m0[e] = (_Float16)0.f; m1[e] = (_Float16)0.f; }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const _Float16* cp = &conv[((2 * ppy + dy) * ST_CW + 2 * ppx + dx) * 64 + cg];
        const f16x8 a = *reinterpret_cast<const f16x8*>(cp), b = *reinterpret_cast<const f16x8*>(cp + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { m0[e] = a[e] > m0[e] ? a[e] : m0[e]; m1[e] = b[e] > m1[e] ? b[e] : m1[e]; }
      }
    uint4* yp = y + ((((long long)t * PH + py) * PW + px) * 64 + cg) / 8;
    yp[0] = __builtin_bit_cast(uint4, m0); yp[1] = __builtin_bit_cast(uint4, m1);
  }
  __syncthreads();                                                     // the next tile's patch / conv tile overwrite this one's
  }
}

}  // namespace

// Stem (7x7 / stride 2 conv with the kernel padded to 7 x 8 taps, FrozenBN folded, ReLU) + 3x3 / stride 2 max pool of a ResNet in one launch:
// x f32 [T, H, W, 4] (W even), w16 fp16 [64, 7, 8, 4], bias f32 [64] -> y fp16 [T, PH, PW, 64].
extern "C" int ovis_resnet_stem_pool_f16(const float* x, const void* w16, const float* bias, void* y_f16, int T, int H, int W, ovis_stream_t stream) {
  OVIS_REQUIRE(x && w16 && bias && y_f16, "resnet_stem_pool: null pointer");
  OVIS_REQUIRE(T > 0 && H > 0 && W > 0 && W % 2 == 0, "resnet_stem_pool: need an even width (frames are padded to a multiple of 32)");
  OVIS_REQUIRE((((uintptr_t)x | (uintptr_t)w16 | (uintptr_t)bias | (uintptr_t)y_f16) & 15) == 0, "resnet_stem_pool: 16-byte alignment");
  const int OH = (H - 1) / 2 + 1, OW = W / 2, PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
  const int tiles_x = ovis::cdiv(PW, ST_PW), tiles_y = ovis::cdiv(PH, ST_PH);
  OVIS_REQUIRE((long long)T * tiles_x * tiles_y < (1ll << 31), "resnet_stem_pool: too many tiles");
  hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)(T * ovis::cdiv(tiles_x, g_stem_xt) * tiles_y)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const uint4*)w16, bias,
                     (uint4*)y_f16, T, H, W, OH, OW, PH, PW, tiles_x, tiles_y, g_stem_xt);
  return ovis::check_launch("resnet_stem_pool");
}

extern "C" int ovis_stem_tiles(int xt) { g_stem_xt = xt >= 1 ? xt : 2; return OVIS_OK; }   // lab only

extern "C" int ovis_conv_h16(const void* x_f16, const void* w_f16, void* y, int out_f16, int T, int H, int W, int Cin, int Cout, int ksize,
                             int stride, const float* bias, const float* residual, int act, ovis_stream_t stream) {
  OVIS_REQUIRE(x_f16 && w_f16 && y, "conv_h16: null pointer");
  OVIS_REQUIRE(T > 0 && H > 0 && W > 0 && (ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv_h16: 1x1 / 3x3 (pad 1), stride 1 / 2");
  OVIS_REQUIRE(Cin % 64 == 0 && Cin >= 64 && Cout % 64 == 0 && Cout >= 64, "conv_h16: Cin (%d) and Cout (%d) must be multiples of 64", Cin, Cout);
  OVIS_REQUIRE(act >= 0 && act <= 3 && !(out_f16 && residual), "conv_h16: bad activation, or an fp16 output with a residual");
  const int pad = ksize / 2;
  const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
  const long long M = (long long)T * OH * OW;
  OVIS_REQUIRE(OH > 0 && OW > 0 && M < (1ll << 31) && (long long)T * H * W * Cin * 2 + (2ll * W + 2) * Cin * 2 < (1ll << 31),
               "conv_h16: tensor too large for 32-bit byte offsets");
  OVIS_REQUIRE((((uintptr_t)x_f16 | (uintptr_t)w_f16 | (uintptr_t)y) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
               (!residual || ((uintptr_t)residual & 15) == 0), "conv_h16: 16-byte alignment");
  CH16Args p;
  p.X = (const _Float16*)x_f16; p.Wt = (const _Float16*)w_f16; p.Y = y; p.bias = bias; p.R = residual;
  p.zeros = zero_page();
  OVIS_REQUIRE(p.zeros, "conv_h16: cannot allocate the page of zeros");
  p.T = T; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.stride = stride; p.act = act; p.M = (int)M;
  // Tile width and LDS slots, measured on the four 3x3 shapes of ResNet-50 at 5 x 736 x 1280 (profiles/r04/conv_h16_tiles.txt).  The kernel is
  // bound by the L2 -> LDS rate per CU, so what matters is how many workgroups are RESIDENT: three slots = 96 KB (128 columns: one workgroup
  // per CU) / 72 KB (64 columns: two), two slots = 64 / 48 KB (two / three).  Three slots while the whole grid is resident with them (res5,
  // 64-column tiles, 288 workgroups: 58.8 us against 67 with two), two slots as soon as that adds resident workgroups (res4, 128 columns, 288
  // workgroups: 46.7 us against 69 -- with three slots the 32 workgroups of a second round run alone); 64-column tiles where 128-column
  // ones would leave CUs empty (res5: 144 workgroups, 62.8 us).
  int bn = (Cout % 128 == 0) ? 128 : 64;
  if (bn == 128 && ovis::cdiv(M, 128) * (Cout / 128) < 256) bn = 64;
  if (g_ch_bn == 64 || (g_ch_bn == 128 && Cout % 128 == 0)) bn = g_ch_bn;
  p.tiles_n = Cout / bn;
  const unsigned grid = ovis::cdiv(M, 128) * (unsigned)p.tiles_n;
  const int nst = g_ch_nst ? g_ch_nst : (grid <= 256u * (bn == 128 ? 1u : 2u) ? 3 : 2);
  hipStream_t s = (hipStream_t)stream;
#define CH_LAUNCH(BN_, TAPS_, O16_) do { if (nst == 2) hipLaunchKernelGGL((conv_h16_kernel<BN_, TAPS_, O16_, 2>), dim3(grid), dim3(256), 0, s, p); \
                                         else hipLaunchKernelGGL((conv_h16_kernel<BN_, TAPS_, O16_, 3>), dim3(grid), dim3(256), 0, s, p); } while (0)
  if (g_ch_dbg && ksize == 3 && out_f16 && nst == 2) {              // lab: DMA-only / MFMA-only variants of the 3x3 kernel
    if (bn == 128) { if (g_ch_dbg == 1) hipLaunchKernelGGL((conv_h16_kernel<128, 9, true, 2, 1>), dim3(grid), dim3(256), 0, s, p);
                     else hipLaunchKernelGGL((conv_h16_kernel<128, 9, true, 2, 2>), dim3(grid), dim3(256), 0, s, p); }
    else { if (g_ch_dbg == 1) hipLaunchKernelGGL((conv_h16_kernel<64, 9, true, 2, 1>), dim3(grid), dim3(256), 0, s, p);
           else hipLaunchKernelGGL((conv_h16_kernel<64, 9, true, 2, 2>), dim3(grid), dim3(256), 0, s, p); }
  } else if (ksize == 3) {
    if (bn == 128) { if (out_f16) CH_LAUNCH(128, 9, true); else CH_LAUNCH(128, 9, false); }
    else { if (out_f16) CH_LAUNCH(64, 9, true); else CH_LAUNCH(64, 9, false); }
  } else {
    if (bn == 128) { if (out_f16) CH_LAUNCH(128, 1, true); else CH_LAUNCH(128, 1, false); }
    else { if (out_f16) CH_LAUNCH(64, 1, true); else CH_LAUNCH(64, 1, false); }
  }
#undef CH_LAUNCH
  return ovis::check_launch("conv_h16");
}

extern "C" int ovis_conv_h16_slots(int nst) { g_ch_nst = (nst == 2 || nst == 3) ? nst : 0; return OVIS_OK; }   // lab / tests only: 0 automatic
extern "C" int ovis_conv_h16_debug(int d) { g_ch_dbg = (d == 1 || d == 2) ? d : 0; return OVIS_OK; }   // lab only
extern "C" int ovis_conv_h16_bn(int bn) { g_ch_bn = (bn == 64 || bn == 128) ? bn : 0; return OVIS_OK; }          // lab only

extern "C" int ovis_maxpool3x3s2_nhwc_f16(const void* x, void* y, int N, int H, int W, int C, ovis_stream_t stream) {
  OVIS_REQUIRE(x && y, "maxpool (fp16): null pointer");
  OVIS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "maxpool (fp16): C %% 8 == 0, 16-byte alignment");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_h16_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, N, H,
                     W, C / 8, OH, OW);
  return ovis::check_launch("maxpool (fp16)");
}
