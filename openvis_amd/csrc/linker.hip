// Temporal instance linker of the online models (MinVIS tracker) on the GPU.
//
// Replaces openvis/modeling/minvis.py:28-72 (match_via_embeds / batch_video_match_via_embeds): for t = 0..T-1,
//   cost[i][j] = 1 - <tgt_i/|tgt_i|, cur_j/|cur_j|>,  tgt = embeds[t-1][indices[t-1]] (tgt = embeds[0] for t = 0),
//   indices[t] = column assignment of scipy.optimize.linear_sum_assignment(cost) (rows = targets, cols = current).
// The reference syncs to the host and calls scipy once per frame; here the whole chain over T runs in ONE launch of
// one workgroup (the chain is inherently sequential in T and the 100x100 problems are tiny), so there is no
// GPU->CPU round trip per frame.  The assignment is the Jonker-Volgenant shortest-augmenting-path algorithm in the
// form scipy uses (rectangular_lsap.cpp: dual variables u, v; ties prefer an unassigned column), in f64 like scipy.
// Also here: the row gather that applies the permutation to per-frame tensors (utils/index.py:4-18 batch_index).
#include "common.h"

namespace {

constexpr int LNK_THREADS = 256;

struct ArgMin { double v; int j; int free_col; };

__device__ __forceinline__ bool better(double v, int j, int fr, double bv, int bj, int bfr) {
  // smaller value; ties: a column that is still unassigned first (scipy), then the lower column index
  if (v < bv) return true;
  if (v > bv) return false;
  if (fr != bfr) return fr > bfr;
  return j < bj;
}

__global__ void __launch_bounds__(LNK_THREADS)
hungarian_link_kernel(const float* __restrict__ embeds, int* __restrict__ indices, float* __restrict__ cost_ws,
                      float* __restrict__ norm_ws, int T, int Q, int C) {
  extern __shared__ double sh[];
  double* u = sh;                        // [Q]
  double* v = u + Q;                     // [Q]
  double* spc = v + Q;                   // shortest path costs [Q]
  int* path = reinterpret_cast<int*>(spc + Q);   // [Q]
  int* col4row = path + Q;               // [Q]
  int* row4col = col4row + Q;            // [Q]
  int* in_sr = row4col + Q;              // [Q]
  int* in_sc = in_sr + Q;                // [Q]
  __shared__ double red_v[LNK_THREADS / 64];
  __shared__ int red_j[LNK_THREADS / 64], red_f[LNK_THREADS / 64];
  __shared__ int s_i, s_sink;
  __shared__ double s_min;
  const int tid = threadIdx.x;

  for (int t = 0; t < T; ++t) {
    const float* cur = embeds + (long long)t * Q * C;
    // normalised copies: norm_ws[0:Q*C] = target rows, norm_ws[Q*C:2*Q*C] = current rows
    float* tn = norm_ws;
    float* cn = norm_ws + (long long)Q * C;
    for (int r = tid; r < 2 * Q; r += LNK_THREADS) {
      const bool is_t = r < Q;
      const int q = is_t ? r : r - Q;
      const float* src = is_t ? (t == 0 ? cur + (long long)q * C
                                        : embeds + ((long long)(t - 1) * Q + indices[(t - 1) * Q + q]) * C)
                              : cur + (long long)q * C;
      float ss = 0.f;
      for (int c = 0; c < C; ++c) ss += src[c] * src[c];
      const float nrm = sqrtf(ss);
      float* dst = (is_t ? tn : cn) + (long long)q * C;
      for (int c = 0; c < C; ++c) dst[c] = src[c] / nrm;
    }
    __threadfence_block();
    __syncthreads();
    // cost[i][j] = 1 - <tn_i, cn_j>
    for (int e = tid; e < Q * Q; e += LNK_THREADS) {
      const int i = e / Q, j = e % Q;
      const float* a = tn + (long long)i * C;
      const float* b = cn + (long long)j * C;
      float d = 0.f;
      for (int c = 0; c < C; ++c) d = fmaf(a[c], b[c], d);
      cost_ws[e] = 1.0f - d;
    }
    for (int q = tid; q < Q; q += LNK_THREADS) { u[q] = 0.0; v[q] = 0.0; col4row[q] = -1; row4col[q] = -1; }
    __threadfence_block();
    __syncthreads();

    for (int cur_row = 0; cur_row < Q; ++cur_row) {
      for (int q = tid; q < Q; q += LNK_THREADS) { spc[q] = INFINITY; in_sr[q] = 0; in_sc[q] = 0; }
      if (tid == 0) { s_i = cur_row; s_sink = -1; s_min = 0.0; }
      __syncthreads();
      while (s_sink < 0) {
        const int i = s_i;
        const double min_val = s_min;
        if (tid == 0) in_sr[i] = 1;
        double bv = INFINITY; int bj = -1, bf = 0;
        for (int j = tid; j < Q; j += LNK_THREADS) {
          if (!in_sc[j]) {
            const double r = min_val + (double)cost_ws[i * Q + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            const int fr = row4col[j] < 0;
            if (bj < 0 || better(spc[j], j, fr, bv, bj, bf)) { bv = spc[j]; bj = j; bf = fr; }
          }
        }
        // workgroup arg-min
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double ov = __shfl_xor(bv, o, 64);
          const int oj = __shfl_xor(bj, o, 64), of = __shfl_xor(bf, o, 64);
          if (oj >= 0 && (bj < 0 || better(ov, oj, of, bv, bj, bf))) { bv = ov; bj = oj; bf = of; }
        }
        if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_j[tid >> 6] = bj; red_f[tid >> 6] = bf; }
        __syncthreads();
        if (tid == 0) {
          for (int w = 1; w < LNK_THREADS / 64; ++w)
            if (red_j[w] >= 0 && (bj < 0 || better(red_v[w], red_j[w], red_f[w], bv, bj, bf))) { bv = red_v[w]; bj = red_j[w]; bf = red_f[w]; }
          s_min = bv;
          in_sc[bj] = 1;
          if (row4col[bj] < 0) s_sink = bj; else s_i = row4col[bj];
        }
        __syncthreads();
      }
      // dual update (rectangular_lsap.cpp: u[curRow] += minVal; SR rows; SC cols)
      const double min_val = s_min;
      for (int q = tid; q < Q; q += LNK_THREADS) {
        if (q == cur_row) u[q] += min_val;
        else if (in_sr[q]) u[q] += min_val - spc[col4row[q]];
      }
      __syncthreads();
      for (int q = tid; q < Q; q += LNK_THREADS)
        if (in_sc[q]) v[q] -= min_val - spc[q];
      // augment along the path
      if (tid == 0) {
        int j = s_sink;
        while (true) {
          const int i = path[j];
          row4col[j] = i;
          const int tmp = col4row[i];
          col4row[i] = j;
          j = tmp;
          if (i == cur_row) break;
        }
      }
      __syncthreads();
    }
    for (int q = tid; q < Q; q += LNK_THREADS) indices[t * Q + q] = col4row[q];
    __threadfence_block();
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

extern "C" long long ovis_hungarian_link_workspace_bytes(int Q, int C) {
  return ((long long)Q * Q + 2ll * Q * C) * sizeof(float);
}

extern "C" int ovis_hungarian_link_f32(const float* embeds, int* indices, float* workspace, int T, int Q, int C,
                                       ovis_stream_t stream) {
  OVIS_REQUIRE(embeds && indices && workspace, "hungarian_link: null pointer");
  OVIS_REQUIRE(T > 0 && Q > 0 && C > 0 && Q <= 1024, "hungarian_link: bad sizes (Q <= 1024)");
  const size_t shmem = sizeof(double) * 3 * Q + sizeof(int) * 5 * Q;
  hipLaunchKernelGGL(hungarian_link_kernel, dim3(1), dim3(LNK_THREADS), shmem, (hipStream_t)stream, embeds, indices, workspace,
                     workspace + (long long)Q * Q, T, Q, C);
  return ovis::check_launch("hungarian_link");
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
