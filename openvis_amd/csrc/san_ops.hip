// SideAdapter (SAN / BriVIS) specific HBM-bound stages for gfx950.
//   * front image path: bicubic resize of the raw padded frames to the CLIP resolution + /255 + CLIP normalisation,
//     written directly as the patch-embedding im2col matrix (side_adapter.py:150-153);
//   * attention-bias path: adaptive max-pool of the predicted per-head biases to the CLIP token grid and construction
//     of the additive [Q+1+L, Q+1+L] attention bias (side_adapter.py:237-270);
//   * pixel-decoder feature injection: dst += bilinear_resize(src) (msdeformattn.py:338-344).
#include "common.h"

namespace {

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// F.interpolate(x / 255, (R,R), mode="bicubic") (align_corners=False, A=-0.75, border-clamped taps) on frames that are
// zero-padded from (H,W) to (Hp,Wp); then (v - mean)/std; output = im2col of the patchify conv (see clip_crop_kernel).
__global__ void __launch_bounds__(256)
san_front_kernel(const uint8_t* __restrict__ frames, void* __restrict__ Av, int out_f16, int T, int H, int W, int Hp, int Wp,
                 int R, int ps, long long lda, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)T * R * R;
  if (i >= total) return;
  const int ox = (int)(i % R), oy = (int)((i / R) % R), t = (int)(i / ((long long)R * R));
  const float A = -0.75f;
  const float sy = ((float)oy + 0.5f) * ((float)Hp / (float)R) - 0.5f;
  const float sx = ((float)ox + 0.5f) * ((float)Wp / (float)R) - 0.5f;
  const int iy = (int)floorf(sy), ix = (int)floorf(sx);
  const float ty = sy - (float)iy, tx = sx - (float)ix;
  const float cy[4] = {cc2(ty + 1.f, A), cc1(ty, A), cc1(1.f - ty, A), cc2(2.f - ty, A)};
  const float cx[4] = {cc2(tx + 1.f, A), cc1(tx, A), cc1(1.f - tx, A), cc2(2.f - tx, A)};
  const uint8_t* fp = frames + (long long)t * 3 * H * W;
  const long long plane = (long long)H * W;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = min(max(iy - 1 + a, 0), Hp - 1);
    float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int xx = min(max(ix - 1 + b, 0), Wp - 1);
      const bool in = yy < H && xx < W;                       // zero padding beyond the real frame
      const long long o = (long long)(in ? yy : 0) * W + (in ? xx : 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) row[c] += (in ? (float)fp[c * plane + o] / 255.f : 0.f) * cx[b];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += row[c] * cy[a];
  }
  const float r0 = (acc[0] - m0) / s0, r1 = (acc[1] - m1) / s1, r2 = (acc[2] - m2) / s2;
  const int G = R / ps;
  const long long rowi = (long long)t * G * G + (oy / ps) * G + (ox / ps);
  const int col = (oy % ps) * ps + (ox % ps);
  if (out_f16) {
    _Float16* ap = reinterpret_cast<_Float16*>(Av) + rowi * lda + col;
    ap[0] = (_Float16)r0; ap[ps * ps] = (_Float16)r1; ap[2 * ps * ps] = (_Float16)r2;
  } else {
    float* ap = reinterpret_cast<float*>(Av) + rowi * lda + col;
    ap[0] = r0; ap[ps * ps] = r1; ap[2 * ps * ps] = r2;
  }
}

// F.adaptive_max_pool2d over N planes [H,W] -> [OH,OW]: window rows floor(i*H/OH) .. ceil((i+1)*H/OH)
__global__ void __launch_bounds__(256)
adaptive_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, long long N, int H, int W, int OH, int OW) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = N * OH * OW;
  if (i >= total) return;
  const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
  const long long n = i / ((long long)OW * OH);
  const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
  const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
  const float* p = x + n * H * W;
  float m = -INFINITY;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) m = fmaxf(m, p[(long long)yy * W + xx]);
  y[i] = m;
}

// additive attention bias of the SideAdapter back blocks (side_adapter.py:253-265):
//   tokens = [Q sos | cls | L patches]; out[bn, r, c]:
//     c < Q            : r == c ? 0 : -100          (nobody looks at other sos tokens)
//     r < Q, c == Q    : -100                        (sos does not look at cls)
//     r < Q, c > Q     : pooled[bn, r, c - Q - 1]    (sos -> patches: predicted bias)
//     otherwise        : 0
__global__ void __launch_bounds__(256)
san_bias_kernel(const float* __restrict__ pooled, float* __restrict__ out, long long BN, int Q, int L, int ld) {
  const int S = Q + 1 + L;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = BN * S * ld;
  if (i >= total) return;
  const int c = (int)(i % ld), r = (int)((i / ld) % S);
  const long long bn = i / ((long long)ld * S);
  float v = 0.f;
  if (c < Q) v = (r == c) ? 0.f : -100.f;
  else if (r < Q && c == Q) v = -100.f;
  else if (r < Q && c < S) v = pooled[(bn * Q + r) * L + (c - Q - 1)];
  out[i] = v;
}

// dst[n,y,x,:] += bilinear_resize(src[n], (H,W))[y,x,:]   (align_corners=False), NHWC, C % 4 == 0
__global__ void __launch_bounds__(256)
resize_add_kernel(float* __restrict__ dst, const float* __restrict__ src, int N, int H, int W, int C, int h, int w) {
  const int c4n = C >> 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * H * W * c4n;
  if (i >= total) return;
  const int cq = (int)(i % c4n);
  long long r = i / c4n;
  const int ox = (int)(r % W); r /= W;
  const int oy = (int)(r % H);
  const int n = (int)(r / H);
  const float sy = fmaxf(((float)oy + 0.5f) * ((float)h / (float)H) - 0.5f, 0.f);
  const float sx = fmaxf(((float)ox + 0.5f) * ((float)w / (float)W) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const float4* sb = reinterpret_cast<const float4*>(src + (long long)n * h * w * C);
  const float4 a = sb[((long long)y0 * w + x0) * c4n + cq], b = sb[((long long)y0 * w + x1) * c4n + cq];
  const float4 c = sb[((long long)y1 * w + x0) * c4n + cq], d = sb[((long long)y1 * w + x1) * c4n + cq];
  float4 v = reinterpret_cast<float4*>(dst)[i];
  v.x += hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
  v.y += hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
  v.z += hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
  v.w += hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
  reinterpret_cast<float4*>(dst)[i] = v;
}

}  // namespace

extern "C" int ovis_san_front_patches(const uint8_t* frames, void* A, int out_f16, int T, int H, int W, int Hp, int Wp,
                                      int resolution, int patch, long long lda, const float* mean3_host, const float* std3_host,
                                      ovis_stream_t stream) {
  OVIS_REQUIRE(frames && A && mean3_host && std3_host, "san_front_patches: null pointer");
  OVIS_REQUIRE(T > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && resolution > 0 && patch > 0 && resolution % patch == 0,
               "san_front_patches: bad geometry");
  OVIS_REQUIRE(lda >= 3ll * patch * patch, "san_front_patches: lda smaller than a patch row (3*patch*patch)");
  const long long total = (long long)T * resolution * resolution;
  hipLaunchKernelGGL(san_front_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, frames, A, out_f16, T, H,
                     W, Hp, Wp, resolution, patch, lda, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                     std3_host[2]);
  return ovis::check_launch("san_front_patches");
}

extern "C" int ovis_adaptive_maxpool2d_f32(const float* x, float* y, long long N, int H, int W, int OH, int OW,
                                           ovis_stream_t stream) {
  OVIS_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "adaptive_maxpool2d: bad arguments");
  const long long total = N * OH * OW;
  hipLaunchKernelGGL(adaptive_maxpool_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, OH, OW);
  return ovis::check_launch("adaptive_maxpool2d");
}

extern "C" int ovis_san_attn_bias_f32(const float* pooled, float* out, long long BN, int Q, int L, int ld,
                                      ovis_stream_t stream) {
  OVIS_REQUIRE(pooled && out && BN > 0 && Q > 0 && L > 0 && ld >= Q + 1 + L, "san_attn_bias: bad arguments");
  const long long total = BN * (Q + 1 + L) * ld;
  hipLaunchKernelGGL(san_bias_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, pooled, out, BN, Q, L, ld);
  return ovis::check_launch("san_attn_bias");
}

extern "C" int ovis_bilinear_resize_add_nhwc_f32(float* dst, const float* src, int N, int H, int W, int C, int h, int w,
                                                 ovis_stream_t stream) {
  OVIS_REQUIRE(dst && src && N > 0 && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "bilinear_resize_add: bad arguments");
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(resize_add_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, N, H, W, C, h, w);
  return ovis::check_launch("bilinear_resize_add");
}
