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

}  // namespace

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
