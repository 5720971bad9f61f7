// Temporal instance linker of the online models (MinVIS tracker) on the GPU.
//
// Replaces openvis/modeling/minvis.py:28-72 (match_via_embeds / batch_video_match_via_embeds): for t = 0..T-1,
//   cost[i][j] = 1 - <tgt_i/|tgt_i|, cur_j/|cur_j|>,  tgt = embeds[t-1][indices[t-1]] (tgt = embeds[0] for t = 0),
//   indices[t] = column assignment of scipy.optimize.linear_sum_assignment(cost) (rows = targets, cols = current).
// The reference syncs to the host and calls scipy once per frame; here all frame-to-frame cosine matrices come from ONE
// batched MFMA GEMM; because a row permutation of a cost matrix only permutes its optimal assignment, the T assignment
// problems are solved INDEPENDENTLY (one single-wavefront workgroup per frame, in parallel) on the un-permuted rows and
// the sequential part shrinks to composing T permutations.  No GPU->CPU round trip per frame.  The assignment is the Jonker-Volgenant shortest-augmenting-path algorithm in the
// form scipy uses (rectangular_lsap.cpp: dual variables u, v; ties prefer an unassigned column), in f64 like scipy.
// Also here: the row gather that applies the permutation to per-frame tensors (utils/index.py:4-18 batch_index).
#include "common.h"

namespace {

// ---- stage 1 (parallel): L2-normalise every embedding row ---------------------------------------------------------
__global__ void __launch_bounds__(256)
normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[r * C + c]; ss += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float nrm = sqrtf(ss);
  for (int c = lane; c < C; c += 64) y[r * C + c] = x[r * C + c] / nrm;
}

// ---- stage 3 (sequential chain, ONE wavefront): Jonker-Volgenant per frame on cost = 1 - G_t[prev[i]][j] -----------
// G_t = En_{t-1} En_t^T (G_0 = En_0 En_0^T) comes from one batched MFMA GEMM (stage 2).  A single 64-lane wavefront
// owns the whole chain: lane l owns columns l, l+64, ...; the arg-min over columns is 6 cross-lane shuffles, and the
// only synchronisation is the (single-wave, i.e. free) workgroup barrier that orders LDS traffic.
// minimum of an unsigned over the 64 lanes, the same value in every lane: quad / half-row / row mirrors make every lane of a
// 16-lane row hold its row's minimum, two row broadcasts bring it to lane 63, which is read back as a scalar (DPP: VALU only)
__device__ __forceinline__ unsigned lnk_wave_min(unsigned v) {
  int x = (int)v;
  auto umin = [](int a, int b) { return (int)min((unsigned)a, (unsigned)b); };
  x = umin(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
  x = umin(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
  x = umin(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));    // row_half_mirror
  x = umin(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));    // row_mirror
  x = umin(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));    // row_bcast:15 -> rows 1, 3
  x = umin(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false));    // row_bcast:31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ bool lnk_better(double v, int j, int fr, double bv, int bj, int bfr) {
  // smaller value; ties: a still-unassigned column first (scipy's rule), then the lower column index
  if (v < bv) return true;
  if (v > bv) return false;
  if (fr != bfr) return fr > bfr;
  return j < bj;
}

// NC = columns per lane (Q <= 64 NC); G_LDS: the frame's Q x Q cosine matrix is staged in LDS (Q*Q*4 bytes) so that the
// latency-bound inner loop never waits on L2.
template <int NC, bool G_LDS>
__global__ void __launch_bounds__(64)
hungarian_chain_kernel(const float* __restrict__ G, int ldg, int* __restrict__ indices, int T, int Q, int parallel) {
  extern __shared__ double shd[];
  double* u = shd;                                    // [Q] row duals
  int* path = reinterpret_cast<int*>(u + Q);          // [Q] column -> predecessor row
  int* col4row = path + Q;                            // [Q]
  int* row4col = col4row + Q;                         // [Q]
  int* prev = row4col + Q;                            // [Q] previous frame's assignment (row permutation of G)
  float* Gs = reinterpret_cast<float*>(prev + Q);     // [Q*ldg] (G_LDS only)
  const int lane = threadIdx.x;

  // parallel != 0: workgroup b solves frame b on the UN-permuted rows (see compose_assignments_kernel); else the chain
  const int t_begin = parallel ? blockIdx.x : 0, t_end = parallel ? blockIdx.x + 1 : T;
  for (int t = t_begin; t < t_end; ++t) {
    const float* Gt = G + (long long)t * Q * ldg;
    if (G_LDS) {
      const float4* g4 = reinterpret_cast<const float4*>(Gt);
      for (int e = lane; e < Q * ldg / 4; e += 64) reinterpret_cast<float4*>(Gs)[e] = g4[e];
    }
    const float* Gc = G_LDS ? Gs : Gt;
    double v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0.0;
    for (int q = lane; q < Q; q += 64) {
      u[q] = 0.0; col4row[q] = -1; row4col[q] = -1;
      prev[q] = (t == 0 || parallel) ? q : indices[(t - 1) * Q + q];
    }
    __syncthreads();
    // register mirrors of prev[] (constant for the frame) and row4col[] (changes only in the augment step): the tree walk reads them
    // with v_readlane (the row / column index is wave-uniform) instead of two more dependent LDS round trips per step
    int prev_r[NC], r4c_r[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { const int j = lane + 64 * c; prev_r[c] = j < Q ? prev[j] : 0; r4c_r[c] = -1; }
    auto lane_read = [&](const int (&reg)[NC], int idx) {   // reg[idx >> 6] of lane idx & 63, idx wave-uniform
      const int l = idx & 63, c = idx >> 6;
      int r = __builtin_amdgcn_readlane(reg[0], l);
#pragma unroll
      for (int k = 1; k < NC; ++k) { const int rk = __builtin_amdgcn_readlane(reg[k], l); r = c == k ? rk : r; }
      return r;
    };
    for (int cur_row = 0; cur_row < Q; ++cur_row) {
      unsigned in_sc = 0;                               // bit c: column lane + 64c is in the tree
      double spc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) spc[c] = INFINITY;
      int i = cur_row, sink = -1;
      double min_val = 0.0;
      while (sink < 0) {
        i = __builtin_amdgcn_readfirstlane(i);
        const double ui = u[i];
        const float* crow = Gc + lane_read(prev_r, i) * ldg;
        double bv = INFINITY; int bj = -1, bf = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int j = lane + 64 * c;
          if (j < Q && !((in_sc >> c) & 1u)) {
            const double r = min_val + (double)(1.0f - crow[j]) - ui - v[c];
            if (r < spc[c]) { spc[c] = r; path[j] = i; }
            const int fr = r4c_r[c] < 0;
            if (bj < 0 || lnk_better(spc[c], j, fr, bv, bj, bf)) { bv = spc[c]; bj = j; bf = fr; }
          }
        }
        // wave-wide arg-min of (value, unassigned-first, lower column) WITHOUT LDS shuffles (round 2: a 6-step butterfly of
        // ds_bpermute on {f64, int, int} -- 24 LDS round trips in the dependent chain of every tree step, ~3 ms per 36-frame
        // clip).  The value goes through an order-preserving map to a 64-bit unsigned key; three 32-bit DPP min-reductions give
        // the minimal high word, the minimal low word among those lanes, and the minimal tie key (assigned bit, column)
        // among the lanes that hold the minimal value -- the same total order as lnk_better.
        {
          const unsigned long long kb = (unsigned long long)__double_as_longlong(bv + 0.0);             // (-0 -> +0)
          const unsigned long long key = bj < 0 ? ~0ull : (kb ^ ((kb >> 63) ? ~0ull : 0x8000000000000000ull));
          const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
          const unsigned mhi = lnk_wave_min(hi);
          const unsigned mlo = lnk_wave_min(hi == mhi ? lo : 0xffffffffu);
          const bool cand = bj >= 0 && hi == mhi && lo == mlo;
          const unsigned mtk = lnk_wave_min(cand ? ((bf ? 0u : 1u) << 16) | (unsigned)bj : 0xffffffffu);
          const unsigned long long mkey = ((unsigned long long)mhi << 32) | mlo;
          bv = __longlong_as_double((long long)(mkey ^ ((mkey >> 63) ? 0x8000000000000000ull : ~0ull)));
          bj = (int)(mtk & 0xffffu); bf = (mtk >> 16) ? 0 : 1;
        }
        min_val = bv;
        if ((bj & 63) == lane) in_sc |= 1u << (bj >> 6);
        const int r4c = lane_read(r4c_r, __builtin_amdgcn_readfirstlane(bj));
        if (r4c < 0) sink = bj; else i = r4c;
      }
      // dual update, column-wise: the rows of the tree are cur_row and row4col[j] of every scanned column j, so
      //   v[j] -= min_val - spc[j],  u[row4col[j]] += min_val - spc[j]  (sink: spc == min_val, no row),  u[cur_row] += min_val
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int j = lane + 64 * c;
        if (j < Q && ((in_sc >> c) & 1u)) {
          const double d = min_val - spc[c];
          v[c] -= d;
          const int r4c = r4c_r[c];
          if (r4c >= 0 && j != sink) u[r4c] += d;
        }
      }
      if (lane == 0) u[cur_row] += min_val;
      __syncthreads();
      if (lane == 0) {                                  // augment along the alternating path
        int j = sink;
        while (true) {
          const int pi = path[j];
          row4col[j] = pi;
          const int tmp = col4row[pi];
          col4row[pi] = j;
          j = tmp;
          if (pi == cur_row) break;
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < NC; ++c) { const int j = lane + 64 * c; if (j < Q) r4c_r[c] = row4col[j]; }
    }
    for (int q = lane; q < Q; q += 64) indices[t * Q + q] = col4row[q];
    __threadfence();
    __syncthreads();
  }
}

// A row permutation of the cost matrix only permutes the optimal assignment: cost_t = P_{t-1} C_t with
// C_t[r][j] = 1 - <e_{t-1,r}, e_{t,j}>, so  assignment_t[i] = sol_t[indices[t-1][i]]  where sol_t solves C_t on the original row
// order.  The T assignment problems are therefore independent (one workgroup each) and only this trivial composition is
// sequential in t.  (Identical to the sequential scipy chain whenever every optimum is unique, i.e. for any real data.)
__global__ void __launch_bounds__(256)
compose_assignments_kernel(int* __restrict__ indices, int T, int Q) {
  extern __shared__ int cur[];
  for (int q = threadIdx.x; q < Q; q += blockDim.x) cur[q] = indices[q];          // t = 0: sol_0 as is
  __syncthreads();
  for (int t = 1; t < T; ++t) {
    int* row = indices + (long long)t * Q;
    int v[4];
    int n = 0;
    for (int q = threadIdx.x; q < Q; q += blockDim.x) v[n++] = row[cur[q]];        // sol_t[indices[t-1][q]]  (Q <= 1024)
    __syncthreads();
    n = 0;
    for (int q = threadIdx.x; q < Q; q += blockDim.x) { row[q] = v[n]; cur[q] = v[n]; ++n; }
    __syncthreads();
  }
}

// out[(b, m), :] = src[(b, idx[b, m]), :]   rows of `len` floats; element (b,n) of src at src + b*src_bs + n*src_rs
__global__ void __launch_bounds__(256)
batch_index_rows_kernel(const float* __restrict__ src, long long src_bs, long long src_rs, const int* __restrict__ idx,
                        float* __restrict__ out, long long out_bs, long long out_rs, int Bn, int Mn, long long len4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Bn * Mn * len4;
  if (i >= total) return;
  const long long c = i % len4;
  const long long r = i / len4;
  const int m = (int)(r % Mn), b = (int)(r / Mn);
  const int n = idx[b * Mn + m];
  reinterpret_cast<float4*>(out + b * out_bs + m * out_rs)[c] = reinterpret_cast<const float4*>(src + b * src_bs + n * src_rs)[c];
}

}  // namespace

extern "C" int ovis_gemm_nt_f32_batched(const float* A, long long lda, long long a_bs, const float* B, long long ldb,
                                        long long b_bs, float* C, long long ldc, long long c_bs, int batch, int M, int N,
                                        int K, const float* bias, int act, ovis_stream_t stream);

extern "C" long long ovis_hungarian_link_workspace_bytes(int T, int Q, int C) {
  const long long qp = (Q + 3) / 4 * 4;                       // G rows padded to a multiple of 4 floats
  return ((long long)T * Q * C + (long long)T * Q * qp) * sizeof(float);
}

extern "C" int ovis_hungarian_link_f32(const float* embeds, int* indices, float* workspace, int T, int Q, int C,
                                       ovis_stream_t stream) {
  OVIS_REQUIRE(embeds && indices && workspace, "hungarian_link: null pointer");
  OVIS_REQUIRE(T > 0 && Q > 0 && C > 0 && Q <= 1024 && C % 4 == 0, "hungarian_link: need Q <= 1024 and C %% 4 == 0");
  hipStream_t s = (hipStream_t)stream;
  float* En = workspace;                                      // [T,Q,C] unit rows
  float* G = workspace + (long long)T * Q * C;                // [T,Q,ldg]: G_t = En_{t-1} En_t^T, G_0 = En_0 En_0^T
  const int ldg = (Q + 3) / 4 * 4;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(ovis::cdiv((long long)T * Q, 4)), dim3(256), 0, s, embeds, En, (long long)T * Q, C);
  int rc = ovis::check_launch("hungarian_link normalize");
  if (rc) return rc;
  rc = ovis_gemm_nt_f32_batched(En, C, 0, En, C, 0, G, ldg, 0, 1, Q, Q, C, nullptr, 0, stream);
  if (rc) return rc;
  if (T > 1) {
    rc = ovis_gemm_nt_f32_batched(En, C, (long long)Q * C, En + (long long)Q * C, C, (long long)Q * C, G + (long long)Q * ldg, ldg,
                                  (long long)Q * ldg, T - 1, Q, Q, C, nullptr, 0, stream);
    if (rc) return rc;
  }
  const size_t base = sizeof(double) * Q + sizeof(int) * 4 * Q;
  if (Q <= 112)        // 112*112*4 = 50 KB of LDS for the staged cosine matrix
    hipLaunchKernelGGL((hungarian_chain_kernel<2, true>), dim3(T), dim3(64), base + sizeof(float) * Q * ldg, s, G, ldg, indices, T, Q, 1);
  else if (Q <= 256)
    hipLaunchKernelGGL((hungarian_chain_kernel<4, false>), dim3(T), dim3(64), base, s, G, ldg, indices, T, Q, 1);
  else
    hipLaunchKernelGGL((hungarian_chain_kernel<16, false>), dim3(T), dim3(64), base, s, G, ldg, indices, T, Q, 1);
  rc = ovis::check_launch("hungarian_link solve");
  if (rc) return rc;
  hipLaunchKernelGGL(compose_assignments_kernel, dim3(1), dim3(256), sizeof(int) * Q, s, indices, T, Q);
  return ovis::check_launch("hungarian_link compose");
}

extern "C" int ovis_batch_index_rows_f32(const float* src, long long src_bs, long long src_rs, const int* idx, float* out,
                                         long long out_bs, long long out_rs, int B, int M, long long len,
                                         ovis_stream_t stream) {
  OVIS_REQUIRE(src && idx && out, "batch_index_rows: null pointer");
  OVIS_REQUIRE(B > 0 && M > 0 && len > 0 && len % 4 == 0 && src_bs % 4 == 0 && src_rs % 4 == 0 && out_bs % 4 == 0 && out_rs % 4 == 0,
               "batch_index_rows: len and strides must be multiples of 4 floats");
  const long long total = (long long)B * M * (len / 4);
  hipLaunchKernelGGL(batch_index_rows_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, src_bs, src_rs,
                     idx, out, out_bs, out_rs, B, M, len / 4);
  return ovis::check_launch("batch_index_rows");
}
