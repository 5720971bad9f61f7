// HBM-bound helpers of the path for gfx950: pre-processing, pooling, LayerNorm, GroupNorm (NHWC),
// broadcast add, sine position encodings.  All are one-pass-per-element kernels with 16-byte
// accesses; none is reshaped into a GEMM.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- A1: (x - mean)/std, zero pad to [T,Hp,Wp,4] NHWC (4th channel = 0) ---------------------
// openvis/openvis.py:57-62 + detectron2 ImageList.from_tensors(size_divisibility=32).
// One workgroup = 256 consecutive pixels of one padded row (blockIdx.y = t * Hp + y): the frame / row split is one scalar division
// per workgroup (the flat-index form spent two 64-bit divisions per pixel: 307 us per 5-frame 720p clip, 0.3 TB/s).
__global__ void __launch_bounds__(256)
preprocess_kernel(const uint8_t* __restrict__ frames, float* __restrict__ out, int T, int H, int W, int Hp,
                  int Wp, float m0, float m1, float m2, float s0, float s1, float s2) {
  // rows on gridDim.x (limit 2^31 - 1), 256-column chunks on gridDim.y (limit 65 535): a whole video is pre-processed before it is
  // windowed (openvis.py:109, san.py:62), so T * Hp exceeds 65 535 from ~90 frames of 720p on
  const int x = blockIdx.y * 256 + threadIdx.x;
  if (x >= Wp) return;
  const int row = blockIdx.x;                               // t * Hp + y
  const int t = row / Hp, y = row - t * Hp;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < H && x < W) {
    const uint8_t* p = frames + ((long long)t * 3 * H + y) * W + x;
    const long long plane = (long long)H * W;
    v.x = ((float)p[0] - m0) / s0;
    v.y = ((float)p[plane] - m1) / s1;
    v.z = ((float)p[2 * plane] - m2) / s2;
  }
  reinterpret_cast<float4*>(out)[(long long)row * Wp + x] = v;
}

// ---- 3x3 / stride 2 / pad 1 max pool, NHWC (detectron2 BasicStem: F.max_pool2d(x, 3, 2, 1)) ---
__global__ void __launch_bounds__(256)
maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int OH,
                    int OW) {
  const int c4n = C >> 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * OH * OW * c4n;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  long long r = i / c4n;
  const int ow = (int)(r % OW);
  r /= OW;
  const int oh = (int)(r % OH);
  const int n = (int)(r / OH);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if (ih < 0 || ih >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if (iw < 0 || iw >= W) continue;
      const float4 v = reinterpret_cast<const float4*>(x + (((long long)n * H + ih) * W + iw) * C)[c4];
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  reinterpret_cast<float4*>(y)[i] = m;
}

// ---- LayerNorm over the last dim (C % 4 == 0, C <= 4096): one wavefront per row -----------------
// y = LN(x + residual) * gamma + beta ; torch.nn.LayerNorm eps inside the sqrt, biased variance.
// IN16: x is fp16 (the fp16 residual stream of the CLIP tower, the reference's GPU dtype; statistics in f32 as CLIP's LayerNorm
// subclass computes them, model.py:157-163); no residual input in that mode.
template <int MAXV, bool OUT16, bool IN16 = false>
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                 const float* __restrict__ beta, void* __restrict__ yv, long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = C >> 2;
  const float4* xp = reinterpret_cast<const float4*>(x + row * C);
  const uint2* xh = reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(x) + row * C);
  const float4* rp = res ? reinterpret_cast<const float4*>(res + row * C) : nullptr;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < nv) {
      if constexpr (IN16) {
        union { uint2 u; _Float16 h[4]; } pk;
        pk.u = xh[idx];
        v[i] = make_float4((float)pk.h[0], (float)pk.h[1], (float)pk.h[2], (float)pk.h[3]);
      } else {
        v[i] = xp[idx];
      }
      if (!IN16 && rp) { const float4 r = rp[idx]; v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w; }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  float4* yp = reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + row * C);
  uint2* yh = reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(yv) + row * C);
  const float4* gp = reinterpret_cast<const float4*>(gamma);
  const float4* bp = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4 g = gp[idx], b = bp[idx];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if constexpr (OUT16) {
        union { _Float16 h[4]; uint2 u; } pk;
        pk.h[0] = (_Float16)o.x; pk.h[1] = (_Float16)o.y; pk.h[2] = (_Float16)o.z; pk.h[3] = (_Float16)o.w;
        yh[idx] = pk.u;
      } else {
        yp[idx] = o;
      }
    }
  }
}

// fp16 rows in, fp16 rows out with 16-byte accesses (8 halves per lane and step): the LayerNorms of the CLIP tower's fp16 residual
// stream.  One wavefront per row, C % 8 == 0, C <= 1024; statistics and affine in f32.  (The generic kernel above moves 8 bytes per
// lane and instruction on this path: 4.0 TB/s against 5.5 for its f32-input form.)
template <int NV8>
__global__ void __launch_bounds__(256)
layernorm_h16_kernel(const _Float16* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                     _Float16* __restrict__ y, long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n8 = C >> 3;
  const uint4* xp = reinterpret_cast<const uint4*>(x + row * C);
  float v[NV8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int idx = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    if (idx < n8) {
      union { uint4 u; _Float16 h[8]; } pk;
      pk.u = xp[idx];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)pk.h[e]; }
      s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    if (lane + i * 64 < n8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  uint4* yp = reinterpret_cast<uint4*>(y + row * C);
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int idx = lane + i * 64;
    if (idx < n8) {
      const float4 g0 = reinterpret_cast<const float4*>(gamma)[2 * idx], g1 = reinterpret_cast<const float4*>(gamma)[2 * idx + 1];
      const float4 b0 = reinterpret_cast<const float4*>(beta)[2 * idx], b1 = reinterpret_cast<const float4*>(beta)[2 * idx + 1];
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      union { _Float16 h[8]; uint4 u; } o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.h[e] = (_Float16)((v[i][e] - mean) * rstd * g[e] + b[e]);
      yp[idx] = o.u;
    }
  }
}

// (mean, rstd) of fp16 rows, the statistics half of layernorm_h16_kernel (same loads, same two-pass arithmetic, 8 bytes written per
// row instead of the normalised row): input of the LayerNorm-folded GEMM (gemm_f16_pp.hip, LNF).
template <int NV8>
__global__ void __launch_bounds__(256)
row_stats_h16_kernel(const _Float16* __restrict__ x, float2* __restrict__ stats, long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n8 = C >> 3;
  const uint4* xp = reinterpret_cast<const uint4*>(x + row * C);
  float v[NV8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int idx = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    if (idx < n8) {
      using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
      union { u32x4 u; _Float16 h[8]; } pk;
      pk.u = *reinterpret_cast<const u32x4*>(xp + idx);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)pk.h[e]; }
      s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    if (lane + i * 64 < n8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) stats[row] = make_float2(mean, rstd);
}

// (mean, rstd) from the partial (sum, sum of squares) pairs the fp16-residual GEMM's epilogue wrote (gemm_f16_pp.hip, PSTAT): one
// thread per row, slots summed in index order.
__global__ void __launch_bounds__(256)
row_stats_finalize_kernel(const float2* __restrict__ part, int slots, float2* __restrict__ stats, long long rows, float inv_c, float eps) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float s = 0.f, q = 0.f;
  for (int i = 0; i < slots; ++i) { const float2 v = part[row * slots + i]; s += v.x; q += v.y; }
  const float mean = s * inv_c;
  const float var = fmaxf(q * inv_c - mean * mean, 0.f);
  stats[row] = make_float2(mean, 1.f / sqrtf(var + eps));
}

// ---- GroupNorm on NHWC (C % 4 == 0, (C/G) % 4 == 0) -------------------------------------------
// pass 1: per-(n,g) sum / sum of squares accumulated in f64 (block partials -> f64 atomics);
// pass 2: normalise (+ optional bilinear x2-upsampled addend, + optional ReLU).
__global__ void __launch_bounds__(256)
gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int HW, int C, int G, int pix_per_blk) {
  // grid: (chunks, N). thread -> channel quad cq = tid % (C/4), pixel lane pl = tid / (C/4)
  const int c4n = C >> 2;
  const int n = blockIdx.y;
  const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n, npl = blockDim.x / c4n;
  const int p0 = blockIdx.x * pix_per_blk;
  const int p1 = min(HW, p0 + pix_per_blk);
  double s = 0.0, q = 0.0;
  for (int p = p0 + pl; p < p1; p += npl) {
    const float4 v = reinterpret_cast<const float4*>(x + ((long long)n * HW + p) * C)[cq];
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  const int cpg4 = (C / G) >> 2;           // channel quads per group
  const int g = cq / cpg4;
  extern __shared__ double sh[];            // [G][2]
  for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) sh[i] = 0.0;
  __syncthreads();
  atomicAdd(&sh[2 * g], s);
  atomicAdd(&sh[2 * g + 1], q);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) atomicAdd(&stats[(long long)n * 2 * G + i], sh[i]);
}

// (mean, rstd) of every (n, g) from the f64 sums, once (round 5: every thread of gn_apply recomputed them -- two f64 divisions and a
// square root per channel quad): the float pair overwrites the first of the two doubles it was computed from.  Same expressions: same values.
__global__ void __launch_bounds__(256)
gn_finalize_kernel(double* __restrict__ stats, int NG, double cnt, float eps) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= NG) return;
  const double sum = stats[2 * k], sq = stats[2 * k + 1];
  const double mean_d = sum / cnt;
  const double var_d = sq / cnt - mean_d * mean_d;
  const float mean = (float)mean_d;
  const float rstd = 1.f / sqrtf((float)var_d + eps);
  reinterpret_cast<float2*>(stats + 2 * k)[0] = make_float2(mean, rstd);
}

// IDX: int when the (padded) map has fewer than 2^31 channel quads (every 720p / 1080p clip): the index arithmetic below is six integer
// divisions per thread, 64-bit ones cost several times the 32-bit ones
template <typename IDX>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats, const float* __restrict__ gamma,
                const float* __restrict__ beta, float* __restrict__ y, int N, int H, int W, int C, int G,
                float eps, int relu, const float* __restrict__ up, int UH, int UW, int pad) {
  // pad = 1: y is [N][H+2][W+2][C] with a ring of zeros (the input layout of ovis_conv3x3_padded_f32_w3): one thread per channel quad of
  // the PADDED map, the ring threads store zeros -- no memset pass, no state between calls
  const int c4n = C >> 2;
  IDX i = (IDX)blockIdx.x * (IDX)blockDim.x + (IDX)threadIdx.x;
  IDX oidx = i;
  if (pad) {
    const IDX totalp = (IDX)N * (H + 2) * (W + 2) * c4n;
    if (i >= totalp) return;
    const int cqp = (int)(i % c4n);
    const IDX pp = i / c4n;
    const int xx = (int)(pp % (W + 2));
    const IDX rr = pp / (W + 2);
    const int yy = (int)(rr % (H + 2)), nn = (int)(rr / (H + 2));
    if (xx == 0 || yy == 0 || xx == W + 1 || yy == H + 1) { reinterpret_cast<float4*>(y)[i] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    i = (((IDX)nn * H + (yy - 1)) * W + (xx - 1)) * c4n + cqp;
  }
  const IDX total = (IDX)N * H * W * c4n;
  if (i >= total) return;
  const int cq = (int)(i % c4n);
  const IDX pix = i / c4n;
  const int n = (int)(pix / ((IDX)H * W));
  const int g = (cq * 4) / (C / G);
  const float2 mr = reinterpret_cast<const float2*>(stats + ((long long)n * G + g) * 2)[0];      // gn_finalize_kernel
  const float mean = mr.x, rstd = mr.y;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[cq], be = reinterpret_cast<const float4*>(beta)[cq];
  float4 o;
  o.x = (v.x - mean) * rstd * ga.x + be.x;
  o.y = (v.y - mean) * rstd * ga.y + be.y;
  o.z = (v.z - mean) * rstd * ga.z + be.z;
  o.w = (v.w - mean) * rstd * ga.w + be.w;
  if (up) {
    // F.interpolate(up, size=(H,W), mode="bilinear", align_corners=False) added to the normalised map
    const int rem = (int)(pix % ((IDX)H * W));
    const int oy = rem / W, ox = rem % W;
    const float sy = fmaxf(((float)oy + 0.5f) * ((float)UH / (float)H) - 0.5f, 0.f);
    const float sx = fmaxf(((float)ox + 0.5f) * ((float)UW / (float)W) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < UH - 1 ? 1 : 0), x1 = x0 + (x0 < UW - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float4* ub = reinterpret_cast<const float4*>(up + (long long)n * UH * UW * C);
    const float4 a = ub[((long long)y0 * UW + x0) * c4n + cq], b = ub[((long long)y0 * UW + x1) * c4n + cq];
    const float4 c = ub[((long long)y1 * UW + x0) * c4n + cq], d = ub[((long long)y1 * UW + x1) * c4n + cq];
    o.x += hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
    o.y += hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
    o.z += hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
    o.w += hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
  }
  if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
  reinterpret_cast<float4*>(y)[oidx] = o;
}

// out[i] = a[i] + b[i % nb]   (float4 granularity)
__global__ void __launch_bounds__(256)
add_bcast_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out,
                 long long n4, long long nb4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 x = a[i], y = b[i % nb4];
  out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// Sine position encodings (input independent; evaluated once per shape and cached by the host).
// 2-D: pixel_decoder/position_encoding.py:29-53 -> out[H,W,2*npf] (y half | x half), + optional level embed.
// 3-D: transformer_decoder/position_encoding.py:135-165 -> out[T,H,W,2*npf] = cat(y,x) + z.
__global__ void __launch_bounds__(256)
pe_sine_kernel(float* __restrict__ out, int T, int H, int W, int npf, int three_d, const float* __restrict__ add) {
  const int C = 2 * npf;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)T * H * W * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  long long r = i / C;
  const int x = (int)(r % W);
  r /= W;
  const int y = (int)(r % H);
  const int t = (int)(r / H);
  const float scale = 6.283185307179586f, eps = 1e-6f;
  const bool is_y = c < npf;
  const int k = is_y ? c : c - npf;
  const float e = is_y ? (float)(y + 1) / ((float)H + eps) * scale : (float)(x + 1) / ((float)W + eps) * scale;
  const float dim_t = powf(10000.f, 2.f * (float)(k / 2) / (float)npf);
  const float a = e / dim_t;
  float v = (k & 1) ? cosf(a) : sinf(a);
  if (three_d) {
    const float ez = (float)(t + 1) / ((float)T + eps) * scale;
    const float dz = powf(10000.f, 2.f * (float)(c / 2) / (float)C);
    const float az = ez / dz;
    v += (c & 1) ? cosf(az) : sinf(az);
  }
  if (add) v += add[c];
  out[i] = v;
}

// y[i] = mean_t x[t*len + i]   (prompt-ensemble mean over templates, adapter.py:133; sequential sum in t)
// Mask prompt (mask_adapted_clip/model.py:334-338, 349-352): patch tokens whose pooled mask is 0 are REPLACED by the
// learned mask embedding of that depth: x[m, first + l, :] = open[m*L + l] ? x : emb[(emb_rows == 1 ? 0 : l), :].
// One thread per float4; closed tokens are rare (patches outside the frame), so most threads only read one byte.
__global__ void __launch_bounds__(256)
mask_prompt_select_kernel(float4* __restrict__ x, const unsigned char* __restrict__ open, const float4* __restrict__ emb,
                          long long n4, int L, int C4, int tokens_per_item, int first, int emb_rows) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % C4);
  const long long tok = i / C4;                                  // m * L + l
  if (open[tok]) return;
  const int l = (int)(tok % L);
  const long long m = tok / L;
  x[(m * tokens_per_item + first + l) * C4 + c] = emb[(long long)(emb_rows == 1 ? 0 : l) * C4 + c];
}

__global__ void __launch_bounds__(256)
mean_dim0_kernel(const float* __restrict__ x, float* __restrict__ y, int n, long long len) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  float s = 0.f;
  for (int t = 0; t < n; ++t) s += x[(long long)t * len + i];
  y[i] = s / (float)n;
}

}  // namespace

extern "C" int ovis_preprocess_u8_nhwc4(const uint8_t* frames, float* out, int T, int H, int W, int Hp, int Wp,
                                        const float* mean3_host, const float* std3_host, ovis_stream_t stream) {
  OVIS_REQUIRE(frames && out && mean3_host && std3_host, "preprocess: null pointer");
  OVIS_REQUIRE(T > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, "preprocess: bad geometry");
  OVIS_REQUIRE((long long)T * Hp <= 0x7fffffffll && ovis::cdiv(Wp, 256) <= 65535u, "preprocess: too many rows / columns for one launch");
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)(T * Hp), ovis::cdiv(Wp, 256)), dim3(256), 0, (hipStream_t)stream, frames, out,
                     T, H, W, Hp, Wp, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                     std3_host[2]);
  return ovis::check_launch("preprocess");
}

extern "C" int ovis_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C,
                                          ovis_stream_t stream) {
  OVIS_REQUIRE(x && y, "maxpool: null pointer");
  OVIS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad geometry (C %% 4 != 0?)");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H,
                     W, C, OH, OW);
  return ovis::check_launch("maxpool");
}

template <bool OUT16>
static int layernorm_launch(const float* x, const float* residual, const float* gamma, const float* beta, void* y,
                            long long rows, int C, float eps, hipStream_t s) {
  OVIS_REQUIRE(x && gamma && beta && y, "layernorm: null pointer");
  OVIS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 4096, "layernorm: C must be a multiple of 4 and <= 4096");
  const unsigned grid = ovis::cdiv(rows, 4);
  const int nv = C / 4;
  if (nv <= 64) hipLaunchKernelGGL((layernorm_kernel<1, OUT16>), dim3(grid), dim3(256), 0, s, x, residual, gamma, beta, y, rows, C, eps);
  else if (nv <= 256) hipLaunchKernelGGL((layernorm_kernel<4, OUT16>), dim3(grid), dim3(256), 0, s, x, residual, gamma, beta, y, rows, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<16, OUT16>), dim3(grid), dim3(256), 0, s, x, residual, gamma, beta, y, rows, C, eps);
  return ovis::check_launch("layernorm");
}

// decoder_norm + the three Linear layers of the mask-embedding MLP (ReLU between them) on the query rows of a masked-attention decoder
// (forward_prediction_heads, video_mask2former_transformer_decoder.py:454-458 with MLP :204-216), as ONE launch: ten times per clip on 100
// (video decoder) or T x 100 (frame decoders) rows, where four launches of ~5 us each are latency, not work.  A workgroup owns R rows: one
// wavefront per row normalises it (layernorm_kernel's two-pass f32 formulas), then thread j owns output column j of every layer for the R
// rows -- the rows sit in LDS (broadcast reads), the weights are read TRANSPOSED ([in][out]: consecutive lanes, consecutive addresses) from
// L2.  f32 FMA chains, four partial sums per output (one per k quarter).
template <int C, int R>
__global__ void __launch_bounds__(C)
ln_mlp3_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ wt0,
               const float* __restrict__ b0, const float* __restrict__ wt1, const float* __restrict__ b1, const float* __restrict__ wt2,
               const float* __restrict__ b2, float* __restrict__ dec, float* __restrict__ out, int rows, float eps) {
  static_assert(C == 256 && R == C / 64, "one wavefront per row, one float4 per lane");
  __shared__ __attribute__((aligned(16))) float buf[2][R][C];
  __shared__ __attribute__((aligned(16))) float red[C / 64][R][C];      // partial sums of the 4 k ranges
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row0 = blockIdx.x * R;
  {
    const int row = min(row0 + w, rows - 1);
    const float4 v = reinterpret_cast<const float4*>(x + (long long)row * C)[lane];
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) / (float)C;
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float rstd = 1.f / sqrtf(wave_sum((a * a + b * b) + (c * c + d * d)) / (float)C + eps);
    const float4 g = reinterpret_cast<const float4*>(gamma)[lane], bt = reinterpret_cast<const float4*>(beta)[lane];
    const float4 y = make_float4(a * rstd * g.x + bt.x, b * rstd * g.y + bt.y, c * rstd * g.z + bt.z, d * rstd * g.w + bt.w);
    reinterpret_cast<float4*>(&buf[0][w][0])[lane] = y;
    if (dec && row0 + w < rows) reinterpret_cast<float4*>(dec + (long long)row * C)[lane] = y;
  }
  __syncthreads();
  // one layer: wavefront w takes the k range [64 w, 64 w + 64), lane l the 4 output columns 4 l .. 4 l + 3 (one 16-byte load per k: a wavefront
  // reads whole 1 KB weight rows) for the R rows -- 64 independent loads per thread, issued 16 at a time; the 4 partial sums of an output meet
  // in LDS and are added in wavefront order.  (Thread j = column j over all 256 k, 4 loads in flight, was latency bound: 26 us per call.)
  auto layer = [&](const float (*in)[C], const float* __restrict__ wt, const float* __restrict__ bias, bool relu, float (*o)[C]) {
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* wp = reinterpret_cast<const float4*>(wt + (long long)(64 * w) * C) + lane;
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      float4 wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = wp[(long long)(kk + u) * (C / 4)];
#pragma unroll
      for (int u4 = 0; u4 < 16; u4 += 4)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float4 xv = *reinterpret_cast<const float4*>(&in[r][64 * w + kk + u4]);
          const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[r].x = __builtin_fmaf(xs[e], wv[u4 + e].x, acc[r].x); acc[r].y = __builtin_fmaf(xs[e], wv[u4 + e].y, acc[r].y);
            acc[r].z = __builtin_fmaf(xs[e], wv[u4 + e].z, acc[r].z); acc[r].w = __builtin_fmaf(xs[e], wv[u4 + e].w, acc[r].w);
          }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) reinterpret_cast<float4*>(&red[w][r][0])[lane] = acc[r];
    __syncthreads();
    const float bj = bias[tid];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v = ((red[0][r][tid] + red[1][r][tid]) + (red[2][r][tid] + red[3][r][tid])) + bj;
      v = relu ? fmaxf(v, 0.f) : v;
      if (o) o[r][tid] = v;
      else if (row0 + r < rows) out[(long long)(row0 + r) * C + tid] = v;
    }
    __syncthreads();
  };
  layer(buf[0], wt0, b0, true, buf[1]);
  layer(buf[1], wt1, b1, true, buf[0]);
  layer(buf[0], wt2, b2, false, nullptr);
}

extern "C" int ovis_ln_mlp3_f32(const float* x, const float* gamma, const float* beta, const float* wt0, const float* b0, const float* wt1,
                                const float* b1, const float* wt2, const float* b2, float* dec, float* out, int rows, int C, float eps,
                                ovis_stream_t stream) {
  OVIS_REQUIRE(x && gamma && beta && wt0 && b0 && wt1 && b1 && wt2 && b2 && out, "ln_mlp3: null pointer");
  OVIS_REQUIRE(rows > 0 && C == 256, "ln_mlp3: C must be 256 (hidden_dim = mask_dim of every reference config)");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                 reinterpret_cast<uintptr_t>(dec)) & 15) == 0, "ln_mlp3: 16-byte alignment");
  hipLaunchKernelGGL((ln_mlp3_kernel<256, 4>), dim3(ovis::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, wt0, b0, wt1, b1, wt2,
                     b2, dec, out, rows, eps);
  return ovis::check_launch("ln_mlp3");
}

extern "C" int ovis_layernorm_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                  float* y, long long rows, int C, float eps, ovis_stream_t stream) {
  return layernorm_launch<false>(x, residual, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
}

template <bool OUT16>
static int layernorm_f16in_launch(const void* x, const float* gamma, const float* beta, void* y, long long rows, int C, float eps, hipStream_t s) {
  OVIS_REQUIRE(x && gamma && beta && y, "layernorm (fp16 input): null pointer");
  OVIS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 1024, "layernorm (fp16 input): C must be a multiple of 4 and <= 1024");
  const unsigned grid = ovis::cdiv(rows, 4);
  const float* xf = reinterpret_cast<const float*>(x);
  if (OUT16 && C % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const _Float16* xh = reinterpret_cast<const _Float16*>(x);
    _Float16* yh = reinterpret_cast<_Float16*>(y);
    if (C <= 512) hipLaunchKernelGGL(layernorm_h16_kernel<1>, dim3(grid), dim3(256), 0, s, xh, gamma, beta, yh, rows, C, eps);
    else hipLaunchKernelGGL(layernorm_h16_kernel<2>, dim3(grid), dim3(256), 0, s, xh, gamma, beta, yh, rows, C, eps);
    return ovis::check_launch("layernorm (fp16 rows)");
  }
  if (C / 4 <= 64) hipLaunchKernelGGL((layernorm_kernel<1, OUT16, true>), dim3(grid), dim3(256), 0, s, xf, nullptr, gamma, beta, y, rows, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<4, OUT16, true>), dim3(grid), dim3(256), 0, s, xf, nullptr, gamma, beta, y, rows, C, eps);
  return ovis::check_launch("layernorm (fp16 input)");
}

extern "C" int ovis_row_stats_f16(const void* x_f16, float* stats, long long rows, int C, float eps, ovis_stream_t stream) {
  OVIS_REQUIRE(x_f16 && stats, "row_stats_f16: null pointer");
  OVIS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 1024, "row_stats_f16: C must be a multiple of 8, <= 1024");
  OVIS_REQUIRE((reinterpret_cast<uintptr_t>(x_f16) & 15) == 0 && (reinterpret_cast<uintptr_t>(stats) & 7) == 0, "row_stats_f16: alignment");
  const unsigned grid = (unsigned)ovis::cdiv(rows, 4);
  const _Float16* xh = reinterpret_cast<const _Float16*>(x_f16);
  hipStream_t s = (hipStream_t)stream;
  if (C <= 512) hipLaunchKernelGGL(row_stats_h16_kernel<1>, dim3(grid), dim3(256), 0, s, xh, reinterpret_cast<float2*>(stats), rows, C, eps);
  else hipLaunchKernelGGL(row_stats_h16_kernel<2>, dim3(grid), dim3(256), 0, s, xh, reinterpret_cast<float2*>(stats), rows, C, eps);
  return ovis::check_launch("row_stats (fp16 rows)");
}

extern "C" int ovis_row_stats_finalize(const float* part, int slots, float* stats, long long rows, int C, float eps, ovis_stream_t stream) {
  OVIS_REQUIRE(part && stats && rows > 0 && slots > 0 && C > 0, "row_stats_finalize: bad arguments");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(stats)) & 7) == 0, "row_stats_finalize: alignment");
  hipLaunchKernelGGL(row_stats_finalize_kernel, dim3((unsigned)ovis::cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(part), slots, reinterpret_cast<float2*>(stats), rows, 1.f / (float)C, eps);
  return ovis::check_launch("row_stats_finalize");
}

extern "C" int ovis_layernorm_f16_to_f16(const void* x_f16, const float* gamma, const float* beta, void* y_f16, long long rows, int C,
                                         float eps, ovis_stream_t stream) {
  return layernorm_f16in_launch<true>(x_f16, gamma, beta, y_f16, rows, C, eps, (hipStream_t)stream);
}

extern "C" int ovis_layernorm_f16_to_f32(const void* x_f16, const float* gamma, const float* beta, float* y, long long rows, int C,
                                         float eps, ovis_stream_t stream) {
  return layernorm_f16in_launch<false>(x_f16, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
}

extern "C" int ovis_layernorm_f32_to_f16(const float* x, const float* residual, const float* gamma, const float* beta,
                                         void* y_f16, long long rows, int C, float eps, ovis_stream_t stream) {
  return layernorm_launch<true>(x, residual, gamma, beta, y_f16, rows, C, eps, (hipStream_t)stream);
}

static int groupnorm_impl(int pad, const float* x, float* y, const float* gamma, const float* beta,
                                       double* stats_ws, int N, int H, int W, int C, int G, float eps, int relu,
                                       const float* up_add, int UH, int UW, ovis_stream_t stream) {
  OVIS_REQUIRE(x && y && gamma && beta && stats_ws, "groupnorm: null pointer");
  OVIS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0 && C <= 1024,
               "groupnorm: need (C/G) %% 4 == 0 and C <= 1024");
  OVIS_REQUIRE(!up_add || (UH > 0 && UW > 0), "groupnorm: bad upsample source size");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * G * N, s);
  if (e != hipSuccess) return ovis::fail(OVIS_ELAUNCH, "groupnorm memset: %s", hipGetErrorString(e));
  const int HW = H * W;
  const int c4n = C / 4;
  const int threads = 256 / c4n > 0 ? (256 / c4n) * c4n : c4n;
  const int pix_per_blk = 128;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(ovis::cdiv(HW, pix_per_blk), N), dim3(threads), sizeof(double) * 2 * G, s, x,
                     stats_ws, HW, C, G, pix_per_blk);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(ovis::cdiv(N * G, 256)), dim3(256), 0, s, stats_ws, N * G, (double)H * W * (C / G), eps);
  const long long total = pad ? (long long)N * (H + 2) * (W + 2) * c4n : (long long)N * HW * c4n;
  if ((long long)N * (H + 2) * (W + 2) * c4n < (1ll << 31) - 256)
    hipLaunchKernelGGL(gn_apply_kernel<int>, dim3(ovis::cdiv(total, 256)), dim3(256), 0, s, x, stats_ws, gamma, beta, y, N, H,
                       W, C, G, eps, relu, up_add, UH, UW, pad);
  else
    hipLaunchKernelGGL(gn_apply_kernel<long long>, dim3(ovis::cdiv(total, 256)), dim3(256), 0, s, x, stats_ws, gamma, beta, y, N, H,
                       W, C, G, eps, relu, up_add, UH, UW, pad);
  return ovis::check_launch("groupnorm");
}

extern "C" int ovis_groupnorm_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta,
                                       double* stats_ws, int N, int H, int W, int C, int G, float eps, int relu,
                                       const float* up_add, int UH, int UW, ovis_stream_t stream) {
  return groupnorm_impl(0, x, y, gamma, beta, stats_ws, N, H, W, C, G, eps, relu, up_add, UH, UW, stream);
}

// the same GroupNorm written into a zero-padded map: y [N][H+2][W+2][C], ring = 0 (input of ovis_conv3x3_padded_f32_w3)
extern "C" int ovis_groupnorm_nhwc_f32_padded(const float* x, float* y_padded, const float* gamma, const float* beta,
                                              double* stats_ws, int N, int H, int W, int C, int G, float eps, int relu,
                                              const float* up_add, int UH, int UW, ovis_stream_t stream) {
  return groupnorm_impl(1, x, y_padded, gamma, beta, stats_ws, N, H, W, C, G, eps, relu, up_add, UH, UW, stream);
}

extern "C" int ovis_add_bcast_f32(const float* a, const float* b, float* out, long long n, long long nb,
                                  ovis_stream_t stream) {
  OVIS_REQUIRE(a && b && out, "add_bcast: null pointer");
  OVIS_REQUIRE(n > 0 && nb > 0 && n % 4 == 0 && nb % 4 == 0 && n % nb == 0, "add_bcast: sizes must be multiples of 4 and nb | n");
  hipLaunchKernelGGL(add_bcast_kernel, dim3(ovis::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                     reinterpret_cast<float4*>(out), n / 4, nb / 4);
  return ovis::check_launch("add_bcast");
}

extern "C" int ovis_pe_sine_f32(float* out, int T, int H, int W, int num_pos_feats, int three_d,
                                const float* add_c, ovis_stream_t stream) {
  OVIS_REQUIRE(out, "pe_sine: null pointer");
  OVIS_REQUIRE(T > 0 && H > 0 && W > 0 && num_pos_feats > 0, "pe_sine: bad geometry");
  const long long total = (long long)T * H * W * 2 * num_pos_feats;
  hipLaunchKernelGGL(pe_sine_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, T, H, W,
                     num_pos_feats, three_d, add_c);
  return ovis::check_launch("pe_sine");
}

extern "C" int ovis_mask_prompt_select_f32(float* x, const unsigned char* patch_open, const float* mask_embedding, int M,
                                           int L, int C, int tokens_per_item, int first_token, int emb_rows,
                                           ovis_stream_t stream) {
  OVIS_REQUIRE(x && patch_open && mask_embedding, "mask_prompt_select: null pointer");
  OVIS_REQUIRE(M > 0 && L > 0 && C > 0 && C % 4 == 0, "mask_prompt_select: C must be a positive multiple of 4");
  OVIS_REQUIRE(first_token >= 0 && first_token + L <= tokens_per_item, "mask_prompt_select: tokens do not fit the item");
  OVIS_REQUIRE(emb_rows == 1 || emb_rows == L, "mask_prompt_select: mask_embedding must have 1 or L rows");
  OVIS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mask_embedding)) & 15) == 0,
               "mask_prompt_select: x / mask_embedding must be 16-byte aligned");
  const long long n4 = (long long)M * L * (C / 4);
  hipLaunchKernelGGL(mask_prompt_select_kernel, dim3(ovis::cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float4*>(x), patch_open, reinterpret_cast<const float4*>(mask_embedding), n4, L, C / 4,
                     tokens_per_item, first_token, emb_rows);
  return ovis::check_launch("mask_prompt_select");
}

extern "C" int ovis_mean_dim0_f32(const float* x, float* y, int n, long long len, ovis_stream_t stream) {
  OVIS_REQUIRE(x && y && n > 0 && len > 0, "mean_dim0: bad arguments");
  hipLaunchKernelGGL(mean_dim0_kernel, dim3(ovis::cdiv(len, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, len);
  return ovis::check_launch("mean_dim0");
}
