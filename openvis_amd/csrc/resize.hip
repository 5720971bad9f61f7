// Test-time input resize on the GPU: detectron2 ResizeShortestEdge -> PIL Image.resize(BILINEAR) of
// the decoded uint8 HWC frame (openvis/data/augmentation.py:368-373; ytvis_dataset_mapper.py:298-313), reproduced
// bit-exactly: Pillow resamples uint8 images with a separable, support-scaled (anti-aliasing) triangle filter in
// 22-bit fixed point -- horizontal pass to an intermediate uint8 image, then the vertical pass, each
// out = clip8((2^21 + sum_k px * coeff_k) >> 22).  The coefficient tables (double arithmetic, normalised, rounded
// to fixed point) are built on the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do
// (openvis_amd/data.py) and passed in; the vertical pass writes the planar CHW layout the model's A1 kernel reads.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[y, ox, c] = resample of src[y, :, c]
__global__ void __launch_bounds__(256)
resize_horizontal_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int H, int W, int OW,
                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)H * OW) return;
  const int ox = (int)(i % OW), y = (int)(i / OW);
  const int xmin = bounds[2 * ox], xn = bounds[2 * ox + 1];
  const int* k = kk + (long long)ox * ksize;
  const uint8_t* row = src + ((long long)y * W + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xn; ++x) {
    const int c = k[x];
    s0 += row[3 * x] * c; s1 += row[3 * x + 1] * c; s2 += row[3 * x + 2] * c;
  }
  uint8_t* o = tmp + i * 3;
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// dst[c, oy, ox] = resample of tmp[:, ox, c]
__global__ void __launch_bounds__(256)
resize_vertical_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst, int H, int OW, int OH,
                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)OH * OW) return;
  const int ox = (int)(i % OW), oy = (int)(i / OW);
  const int ymin = bounds[2 * oy], yn = bounds[2 * oy + 1];
  const int* k = kk + (long long)oy * ksize;
  const uint8_t* col = tmp + ((long long)ymin * OW + ox) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < yn; ++y) {
    const int c = k[y];
    const uint8_t* p = col + (long long)y * OW * 3;
    s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
  }
  const long long plane = (long long)OH * OW;
  dst[i] = clip8(s0); dst[plane + i] = clip8(s1); dst[2 * plane + i] = clip8(s2);
}

// SURVEY.md 8f-2's sketch, round 6: the vertical pass ALSO writes what A1 (preprocess_kernel: openvis.py:57-62) would compute from the frame it has just
// produced -- ((float)v - mean) / std per channel as f32 NHWC4 (channel 3 zero), zero padded to [Hp, Wp] -- so that the resized frame is not read again.
// One thread per pixel of the PADDED image; the uint8 CHW frame is still written: the CLIP crops read it (adapter.py:96-108).  Same expressions as the
// two kernels it replaces: both outputs bit-identical to resize_vertical_kernel + preprocess_kernel.
__global__ void __launch_bounds__(256)
resize_vertical_preprocess_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst, float4* __restrict__ img, int H, int OW, int OH, int Hp,
                                  int Wp, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, float m0, float m1, float m2, float s0_,
                                  float s1_, float s2_) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Hp * Wp) return;
  const int ox = (int)(i % Wp), oy = (int)(i / Wp);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (oy < OH && ox < OW) {
    const int ymin = bounds[2 * oy], yn = bounds[2 * oy + 1];
    const int* k = kk + (long long)oy * ksize;
    const uint8_t* col = tmp + ((long long)ymin * OW + ox) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < yn; ++y) {
      const int c = k[y];
      const uint8_t* p = col + (long long)y * OW * 3;
      s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
    }
    const uint8_t r = clip8(s0), g = clip8(s1), b = clip8(s2);
    const long long plane = (long long)OH * OW, o = (long long)oy * OW + ox;
    dst[o] = r; dst[plane + o] = g; dst[2 * plane + o] = b;
    v.x = ((float)r - m0) / s0_; v.y = ((float)g - m1) / s1_; v.z = ((float)b - m2) / s2_;
  }
  img[i] = v;
}

}  // namespace

extern "C" int ovis_pil_resize_preprocess_u8(const uint8_t* src, int H, int W, uint8_t* tmp, uint8_t* dst, float* img_nhwc4, int OH, int OW, int Hp,
                                             int Wp, const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize,
                                             const float* mean3_host, const float* std3_host, ovis_stream_t stream) {
  OVIS_REQUIRE(src && tmp && dst && img_nhwc4 && xbounds && xk && ybounds && yk && mean3_host && std3_host, "pil_resize_preprocess: null pointer");
  OVIS_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && Hp >= OH && Wp >= OW && xksize > 0 && yksize > 0 && (((uintptr_t)img_nhwc4) & 15) == 0,
               "pil_resize_preprocess: bad sizes (the padded image must hold the resized frame; 16-byte aligned)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(resize_horizontal_kernel, dim3(ovis::cdiv((long long)H * OW, 256)), dim3(256), 0, s, src, tmp, H, W, OW, xbounds,
                     xk, xksize);
  int rc = ovis::check_launch("pil_resize horizontal");
  if (rc) return rc;
  hipLaunchKernelGGL(resize_vertical_preprocess_kernel, dim3(ovis::cdiv((long long)Hp * Wp, 256)), dim3(256), 0, s, tmp, dst,
                     reinterpret_cast<float4*>(img_nhwc4), H, OW, OH, Hp, Wp, ybounds, yk, yksize, mean3_host[0], mean3_host[1], mean3_host[2],
                     std3_host[0], std3_host[1], std3_host[2]);
  return ovis::check_launch("pil_resize vertical + preprocess");
}

extern "C" int ovis_pil_resize_u8_hwc_to_chw(const uint8_t* src, int H, int W, uint8_t* tmp, uint8_t* dst, int OH, int OW,
                                             const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk,
                                             int yksize, ovis_stream_t stream) {
  OVIS_REQUIRE(src && tmp && dst && xbounds && xk && ybounds && yk, "pil_resize: null pointer");
  OVIS_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && xksize > 0 && yksize > 0, "pil_resize: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(resize_horizontal_kernel, dim3(ovis::cdiv((long long)H * OW, 256)), dim3(256), 0, s, src, tmp, H, W, OW, xbounds,
                     xk, xksize);
  int rc = ovis::check_launch("pil_resize horizontal");
  if (rc) return rc;
  hipLaunchKernelGGL(resize_vertical_kernel, dim3(ovis::cdiv((long long)OH * OW, 256)), dim3(256), 0, s, tmp, dst, H, OW, OH, ybounds,
                     yk, yksize);
  return ovis::check_launch("pil_resize vertical");
}
