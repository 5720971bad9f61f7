// Skinny f32 GEMM for the 100-row problems of the masked-attention decoders (queries x Linear; FFNLayer.forward_post,
// video_mask2former_transformer_decoder.py:175-179, and the other nn.Linear modules of the layer loop :413-452).
//
//   C[m,n] = act( sum_k A[m,k] * B[n,k] + bias[n] + R[m,n] ),  M <= 128, K % 256 == 0, f32 everywhere (exact f32 MFMA).
//
// gemm_f32_kernel<64,64> runs these on 8 workgroups with one global round trip per 32-deep K tile: 10 us at K = 256 but 50 us at
// K = 2048 (FFN2 of every decoder layer: 64 serial K tiles on 8 workgroups).  Here the K axis is spread over wavefronts and workgroups:
//   * one workgroup = all 128 rows x 32 columns; its 8 wavefronts take K / 8 each (K = 256: 32 k per wavefront);
//   * a wavefront loads its operand slices straight into the MFMA layout (v_mfma_f32_32x32x2_f32: lane = row, lane half = k;
//     the K order is a free permutation as long as A and B agree, so lane half h owns the 16 consecutive floats [16h, 16h+16)
//     of a 32-wide slice: four 16-byte loads per 32x32 tile) -- every load of the wavefront is in flight before the first MFMA;
//   * the 8 partial 128x32 tiles meet in LDS (128 KB), every thread sums its 8 outputs, applies bias / residual / activation
//     and stores 32 bytes of an output row;
//   * K > 512 (FFN2, K = 2048): grid.y = K / 256 workgroups per column block; partial tiles go to a workspace, the last
//     workgroup to arrive (device-scope counter) sums them in a fixed order -- deterministic, no atomics on the data.
#include "common.h"
#include "gemm_epilogue.h"
#include <map>
#include <mutex>
#include <utility>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int SK_BN = 32, SK_WAVES = 8;
constexpr int SK_MAX_SPLIT = 16;                 // K <= 4096
constexpr int SK_MAX_N = 4096;
constexpr size_t SK_WS_BYTES = (size_t)SK_MAX_SPLIT * 128 * 1024 * 4;   // partial tiles [split][128][N <= 1024]

struct SkArgs {
  const float* A; const float* B; float* C; const float* bias; const float* R;
  long long lda, ldb, ldc, ldr;
  int M, N, K, act;
  int kw;                                        // k range of one wavefront (multiple of 32)
  float* ws; unsigned* counters;                 // split-K only
};

// RT = 32-row tiles per workgroup: 4 (all <= 128 rows; the split-K form of the long-K problems) or 1 (K <= 512: blockIdx.z picks the
// 32-row tile, so a 100 x 256 x 256 problem runs on 32 workgroups instead of 8 and a wavefront issues 16 MFMAs instead of 64)
template <int RT>
__global__ void __launch_bounds__(512)
gemm_f32_skinny_kernel(const SkArgs p) {
  __shared__ __attribute__((aligned(16))) float part[SK_WAVES][RT * 32 * SK_BN];      // RT = 4: 128 KB
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * SK_BN;
  const int split = gridDim.y, ks = blockIdx.y;
  const int k0 = (ks * SK_WAVES + wave) * p.kw;

  const int m0 = RT == 4 ? 0 : blockIdx.z * RT * 32;                 // first row of this workgroup
  f32x16 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int nb = min(n0 + r32, p.N - 1);                                   // clamped: columns >= N are computed and dropped
  const float* bp = p.B + (long long)nb * p.ldb + k0 + 16 * h;
  const float* ap[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) ap[t] = p.A + (long long)min(m0 + t * 32 + r32, p.M - 1) * p.lda + k0 + 16 * h;

  for (int kk = 0; kk < p.kw; kk += 32) {
    float4 a[RT][4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float4*>(bp + kk + 4 * j);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[t][j] = *reinterpret_cast<const float4*>(ap[t] + kk + 4 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bv[4] = {b[j].x, b[j].y, b[j].z, b[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const float av = e == 0 ? a[t][j].x : e == 1 ? a[t][j].y : e == 2 ? a[t][j].z : a[t][j].w;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e], acc[t], 0, 0, 0);
        }
    }
  }

  // partial tiles -> LDS: D[i] of lane l = row 8 (i / 4) + 4 (l / 32) + i % 4, column l % 32
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      part[wave][(t * 32 + 8 * (i >> 2) + 4 * h + (i & 3)) * SK_BN + r32] = acc[t][i];
  __syncthreads();
  if (RT != 4 && tid >= RT * 32 * 4) return;                         // (the wavefronts 2-7 END here: the barriers of the split-K tail count the live ones only)

  // thread -> row ml of the workgroup's rows (m globally), 8 consecutive columns
  const int ml = tid >> 2, c8 = (tid & 3) * 8;
  const int m = m0 + ml;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
  for (int w = 0; w < SK_WAVES; ++w) {
    const float4 x0 = *reinterpret_cast<const float4*>(&part[w][ml * SK_BN + c8]);
    const float4 x1 = *reinterpret_cast<const float4*>(&part[w][ml * SK_BN + c8 + 4]);
    v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
  }

  if (split > 1) {
    // split-K: park the partial tile; the last workgroup of this column block sums the `split` partials in order
    float* wsp = p.ws + ((long long)ks * 128 + m) * p.N + n0 + c8;
    if (m < p.M) {
      if (n0 + c8 + 7 < p.N && (p.N & 3) == 0) {
        *reinterpret_cast<float4*>(wsp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(wsp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (n0 + c8 + e < p.N) wsp[e] = v[e];
      }
    }
    __threadfence();
    __syncthreads();
    const unsigned cidx = blockIdx.z * gridDim.x + blockIdx.x;              // one counter per (row tile, column block)
    if (tid == 0) s_last = atomicAdd(&p.counters[cidx], 1u) == (unsigned)(split - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (m < p.M) {
      const bool vec = n0 + c8 + 7 < p.N && (p.N & 3) == 0;               // (workspace rows are N floats: 16-byte aligned iff N % 4 == 0)
      for (int s0 = 0; s0 < split; s0 += 4) {                                // 8 loads in flight, summed in the fixed order s = 0, 1, ...
        float4 x[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* q = p.ws + ((long long)min(s0 + u, split - 1) * 128 + m) * p.N + n0 + c8;
          if (vec) { x[u][0] = *reinterpret_cast<const float4*>(q); x[u][1] = *reinterpret_cast<const float4*>(q + 4); }
          else {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = n0 + c8 + e < p.N ? q[e] : 0.f;
            x[u][0] = make_float4(t[0], t[1], t[2], t[3]); x[u][1] = make_float4(t[4], t[5], t[6], t[7]);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s0 + u < split) {
            v[0] += x[u][0].x; v[1] += x[u][0].y; v[2] += x[u][0].z; v[3] += x[u][0].w;
            v[4] += x[u][1].x; v[5] += x[u][1].y; v[6] += x[u][1].z; v[7] += x[u][1].w;
          }
      }
    }
    if (tid == 0) p.counters[cidx] = 0;                                     // ready for the next launch on this stream
  }

  if (m >= p.M) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int n = n0 + c8 + e;
    if (n >= p.N) break;
    float x = v[e];
    if (p.bias) x += p.bias[n];
    if (p.R) x += p.R[(long long)m * p.ldr + n];
    if (p.act == 1) x = fmaxf(x, 0.f);
    else if (p.act == 2) x = ovis::quick_gelu(x);
    else if (p.act == 3) x = ovis::gelu_erf(x);
    v[e] = x;
  }
  float* cp = p.C + (long long)m * p.ldc + n0 + c8;
  if (n0 + c8 + 7 < p.N && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {
    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n0 + c8 + e < p.N) cp[e] = v[e];
  }
}

struct SkWorkspace { float* ws; unsigned* counters; };

// split-K workspace (8 MB) + counters per (device, stream): launches on different streams (clips in flight) must not share one;
// allocated at the first split-K launch on that stream and kept for the life of the process
SkWorkspace sk_workspace(hipStream_t stream) {
  static std::map<std::pair<int, hipStream_t>, SkWorkspace> table;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SkWorkspace{nullptr, nullptr};
  std::lock_guard<std::mutex> lock(mu);
  auto it = table.find({dev, stream});
  if (it != table.end()) return it->second;
  void *a = nullptr, *c = nullptr;
  // the counters are zeroed ON THE LAUNCH STREAM: a memset on the null stream is not ordered against a non-blocking stream's first kernel
  if (hipMalloc(&a, SK_WS_BYTES) != hipSuccess || hipMalloc(&c, 4096) != hipSuccess || hipMemsetAsync(c, 0, 4096, stream) != hipSuccess)
    return SkWorkspace{nullptr, nullptr};
  return table[{dev, stream}] = SkWorkspace{(float*)a, (unsigned*)c};
}

int g_skinny = 1, g_skinny_all = 0;

}  // namespace

namespace ovis {

bool gemm_f32_skinny_eligible(const float* A, long long lda, const float* B, long long ldb, int M, int N, int K) {
  if (!g_skinny || M > 128 || M < 1 || K > 256 * SK_MAX_SPLIT || N > SK_MAX_N || N < 8) return false;
  if ((lda & 3) || (ldb & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15)) return false;
  if (K % 256 != 0) return false;                                            // every wavefront a whole number of 32-wide slices
  if (K <= 512 && g_skinny_all < 0) return false;   // lab: -1 = the short-K problems stay on gemm_f32_kernel<64,64> (10 us at K = 256, 8 workgroups)
  if (K > 512 && N > 1024) return false;                                     // split-K workspace bound
  return true;
}

int gemm_f32_skinny_launch(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N, int K,
                           const float* bias, const float* R, long long ldr, int act, hipStream_t stream) {
  SkArgs p{A, B, C, bias, R, lda, ldb, ldc, ldr, M, N, K, act, 0, nullptr, nullptr};
  int split = 1;
  // (K = 2048 without split-K on 32-row workgroups -- lab mode 4 -- is 27 -> 20 us, but its 256-long accumulation chains per wavefront
  // cost the last bit that the C1 mask-bit test of the random-init model resolves: split-K stays)
  if (K > 512 && g_skinny_all != 2) {
    split = K / 256;
    const SkWorkspace w = sk_workspace(stream);
    if (!w.ws) return fail(OVIS_EINVAL, "gemm_nt_f32 (skinny): cannot allocate the split-K workspace");
    p.ws = w.ws; p.counters = w.counters;
  }
  p.kw = K / split / SK_WAVES;
  // K <= 512: one 32-row tile per workgroup (blockIdx.z), no split-K -- 4x the workgroups, a quarter of the MFMAs per wavefront
  // ... and, round 4, the split-K problems too (FFN2 of a decoder layer, 100 x 256 x 2048: 8 x 8 x 4 = 256 workgroups instead of 64; the same
  // per-wavefront k ranges and the same summation order, so the same bits: 32 -> ~12 us)
  if (g_skinny_all <= 0 || (split == 1 && g_skinny_all == 2))
    hipLaunchKernelGGL(gemm_f32_skinny_kernel<1>, dim3(cdiv(N, SK_BN), split, cdiv(M, 32)), dim3(512), 0, stream, p);
  else
    hipLaunchKernelGGL(gemm_f32_skinny_kernel<4>, dim3(cdiv(N, SK_BN), split), dim3(512), 0, stream, p);
  return check_launch("gemm_nt_f32 (skinny)");
}

}  // namespace ovis

// 0: off (gemm_f32_kernel), 1 (default): K <= 512 on 32-row workgroups, long K on the split-K form; 2: K <= 512 on the 128-row form too
// (round-2 layout, tests / lab); 3: K <= 512 stays on gemm_f32_kernel<64,64> (lab A/B)
extern "C" int ovis_set_skinny_gemm(int on) { g_skinny = on ? 1 : 0; g_skinny_all = on == 2 ? 1 : on == 3 ? -1 : on == 4 ? 2 : 0; return OVIS_OK; }   // 4 (lab): no split-K at all
