// K1 — multi-scale deformable attention forward for gfx950 (MI355X).
//
// Replaces the reference's CUDA op (openvis/modeling/pixel_decoder/ops/src/cuda/
// ms_deform_im2col_cuda.cuh:242-304 + :39-89; host side cuda/ms_deform_attn_cuda.cu:25-85).
//
// MI355X design (HBM/L2-bound gather, no matrix work):
//  * The reference maps ONE thread to ONE output channel: a 64-wide wavefront then covers two
//    (query, head) rows and every tap is a 4-byte gather. Here one lane owns VEC=4 consecutive
//    channels, so D/4 lanes cover a head and — for the encoder's M*D = 256 — one wavefront is
//    exactly one query: its 4 taps x 12 points are 128-byte row segments of
//    value[b, idx, m, 0:32] fetched as dwordx4, and its output is one coalesced 1 KiB store.
//  * sampling_loc / attn_weight of a (query, head) are 24 + 12 contiguous floats read as
//    dwordx4 (same address across the D/4 lanes of a head -> one request, broadcast).
//  * L (levels) and P (points) are compile-time for the model's 3x4 configuration so all 16
//    row gathers of a level are in flight together; other (L,P) take the runtime-loop variant.
//  * blockIdx -> query mapping is XCD-aware (common.h:xcd_remap): each XCD's private 4 MiB L2
//    serves a contiguous range of queries, whose sampling windows overlap spatially.
//  * Arithmetic is evaluated in exactly the order of the CUDA kernel, without FMA contraction,
//    so the result is bit-identical to oracle/msda_ref.c.
#include "common.h"

#pragma clang fp contract(off)

namespace {

template <typename T, int VEC> struct VecOf;
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<float, 1> { using type = float; };
template <> struct VecOf<double, 1> { using type = double; };

// 16-bit VALUE storage ("+ bf16-value variant"): the value tensor holds bf16 / fp16, everything else is f32.  The taps are
// widened exactly, the arithmetic is the f32 kernel's -- the result equals ovis_msda_forward_f32 on the widened tensor bit for bit -- and a tap
// is a 64-byte row segment instead of a 128-byte one: half the bytes through the gather path that bounds K1.
struct BF16 { unsigned short b; };
struct FP16 { unsigned short b; };
__device__ __forceinline__ float widen(BF16 v) { return __uint_as_float((unsigned)v.b << 16); }
__device__ __forceinline__ float widen(FP16 v) { return (float)__builtin_bit_cast(_Float16, v.b); }

template <typename T> __device__ __forceinline__ T tfloor(T x);
template <> __device__ __forceinline__ float tfloor<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double tfloor<double>(double x) { return floor(x); }

// BRANCH-FREE tap: the caller passes an address that is always valid (clamped); `pred` only selects value or zero.
// (A predicated load compiles to an exec-masked branch with its own s_waitcnt, which serialises the four gathers of a
// point; per-component selects keep all four in flight.  Results are unchanged bit for bit.)
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(T (&v)[VEC], const T* p, bool pred) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = pred ? t.x : 0.f; v[1] = pred ? t.y : 0.f; v[2] = pred ? t.z : 0.f; v[3] = pred ? t.w : 0.f;
  } else {
    const T t = p[0];
    v[0] = pred ? t : T(0);
  }
}

// four 16-bit values (8 bytes) of a tap, widened to f32
template <typename V16>
__device__ __forceinline__ void load_vec16(float (&v)[4], const V16* p, bool pred) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  const V16 e0{(unsigned short)(t.x & 0xffffu)}, e1{(unsigned short)(t.x >> 16)}, e2{(unsigned short)(t.y & 0xffffu)}, e3{(unsigned short)(t.y >> 16)};
  v[0] = pred ? widen(e0) : 0.f; v[1] = pred ? widen(e1) : 0.f; v[2] = pred ? widen(e2) : 0.f; v[3] = pred ? widen(e3) : 0.f;
}

template <typename V16>
__device__ __forceinline__ void sample_point16(float (&acc)[4], const V16* vp, int H, int W, int qid_stride, float loc_w, float loc_h, float weight) {
  const float h_im = loc_h * (float)H - 0.5f;
  const float w_im = loc_w * (float)W - 0.5f;
  if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - h_low, lw = w_im - w_low;
    const float hh = 1 - lh, hw = 1 - lw;
    const long long h_stride = (long long)W * qid_stride;
    const int yl = h_low < 0 ? 0 : h_low, yh = h_high > H - 1 ? H - 1 : h_high;
    const int xl = w_low < 0 ? 0 : w_low, xh = w_high > W - 1 ? W - 1 : w_high;
    const V16* r0 = vp + yl * h_stride;
    const V16* r1 = vp + yh * h_stride;
    float v1[4], v2[4], v3[4], v4[4];
    load_vec16<V16>(v1, r0 + (long long)xl * qid_stride, h_low >= 0 && w_low >= 0);
    load_vec16<V16>(v2, r0 + (long long)xh * qid_stride, h_low >= 0 && w_high <= W - 1);
    load_vec16<V16>(v3, r1 + (long long)xl * qid_stride, h_high <= H - 1 && w_low >= 0);
    load_vec16<V16>(v4, r1 + (long long)xh * qid_stride, h_high <= H - 1 && w_high <= W - 1);
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float val = (w1 * v1[i] + w2 * v2[i] + w3 * v3[i] + w4 * v4[i]);
      acc[i] += val * weight;
    }
  }
}

// value in 16-bit storage, 4 channels per lane (8-byte taps), runtime levels / points; loc / attw / out f32
template <typename V16>
__global__ void __launch_bounds__(256)
msda_fwd16_kernel(const V16* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                  const float* __restrict__ attw, float* __restrict__ out, long long n_items, int S, int M, int D, int L, int Lq, int P) {
  const unsigned blk = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const long long item = (long long)blk * blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  const int dv = D / 4;
  const int cv = (int)(item % dv);
  const long long sidx = item / dv;
  const int m = (int)(sidx % M);
  const long long b = sidx / ((long long)M * Lq);
  const int qid_stride = M * D;
  const float* lp = loc + sidx * L * P * 2;
  const float* wp = attw + sidx * L * P;
  const V16* vbase = value + b * (long long)S * qid_stride + m * D + cv * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const V16* vp = vbase + lsi[l] * (long long)qid_stride;
    for (int p = 0; p < P; ++p)
      sample_point16<V16>(acc, vp, H, W, qid_stride, lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], wp[l * P + p]);
  }
  *reinterpret_cast<float4*>(out + sidx * D + cv * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <typename V16>
int msda_dispatch16(const void* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attw, float* out, int B, int S,
                    int M, int D, int L, int Lq, int P, hipStream_t stream) {
  OVIS_REQUIRE(value && shapes && lsi && loc && attw && out, "msda_forward (16-bit value): null pointer");
  OVIS_REQUIRE(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, "msda_forward (16-bit value): non-positive dimension");
  OVIS_REQUIRE(D % 4 == 0 && ((uintptr_t)value % 8) == 0 && ((uintptr_t)out % 16) == 0,
               "msda_forward (16-bit value): channels must be a multiple of 4, value 8-byte and out 16-byte aligned");
  const long long n_items = (long long)B * Lq * M * (D / 4);
  hipLaunchKernelGGL((msda_fwd16_kernel<V16>), dim3(ovis::cdiv(n_items, 256)), dim3(256), 0, stream, reinterpret_cast<const V16*>(value), shapes, lsi,
                     loc, attw, out, n_items, S, M, D, L, Lq, P);
  return ovis::check_launch("msda_forward (16-bit value)");
}

// One sampling point: 4 predicated row loads + bilinear blend, accumulated into acc.
template <typename T, int VEC>
__device__ __forceinline__ void sample_point(T (&acc)[VEC], const T* vp, int H, int W, int qid_stride,
                                             T loc_w, T loc_h, T weight) {
  const T h_im = loc_h * (T)H - (T)0.5;
  const T w_im = loc_w * (T)W - (T)0.5;
  if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
    const int h_low = (int)tfloor<T>(h_im), w_low = (int)tfloor<T>(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const T lh = h_im - h_low, lw = w_im - w_low;
    const T hh = 1 - lh, hw = 1 - lw;
    const long long h_stride = (long long)W * qid_stride;
    const int yl = h_low < 0 ? 0 : h_low, yh = h_high > H - 1 ? H - 1 : h_high;       // clamped (always valid) tap coordinates
    const int xl = w_low < 0 ? 0 : w_low, xh = w_high > W - 1 ? W - 1 : w_high;
    const T* r0 = vp + yl * h_stride;
    const T* r1 = vp + yh * h_stride;
    T v1[VEC], v2[VEC], v3[VEC], v4[VEC];
    load_vec<T, VEC>(v1, r0 + (long long)xl * qid_stride, h_low >= 0 && w_low >= 0);
    load_vec<T, VEC>(v2, r0 + (long long)xh * qid_stride, h_low >= 0 && w_high <= W - 1);
    load_vec<T, VEC>(v3, r1 + (long long)xl * qid_stride, h_high <= H - 1 && w_low >= 0);
    load_vec<T, VEC>(v4, r1 + (long long)xh * qid_stride, h_high <= H - 1 && w_high <= W - 1);
    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const T val = (w1 * v1[i] + w2 * v2[i] + w3 * v3[i] + w4 * v4[i]);
      acc[i] += val * weight;
    }
  }
}

// LT/PT > 0: compile-time levels/points (fully unrolled per level); 0: runtime loops.
template <typename T, int VEC, int LT, int PT>
__global__ void __launch_bounds__(256)
msda_fwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                const T* __restrict__ attw, T* __restrict__ out, long long n_items, int S, int M,
                int D, int L_rt, int Lq, int P_rt) {
  const int L = LT > 0 ? LT : L_rt;
  const int P = PT > 0 ? PT : P_rt;
  const unsigned blk = ovis::xcd_remap(blockIdx.x, gridDim.x);
  const long long item = (long long)blk * blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  const int dv = D / VEC;
  const int cv = (int)(item % dv);
  const long long sidx = item / dv;  // (b*Lq + q)*M + m
  const int m = (int)(sidx % M);
  const long long b = sidx / ((long long)M * Lq);
  const int qid_stride = M * D;
  const T* lp = loc + sidx * L * P * 2;
  const T* wp = attw + sidx * L * P;
  const T* vbase = value + b * (long long)S * qid_stride + m * D + cv * VEC;

  T acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0;

  if constexpr (LT > 0 && PT == 4 && VEC == 4) {
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* vp = vbase + lsi[l] * (long long)qid_stride;
      const float4 la = *reinterpret_cast<const float4*>(lp + l * 8);
      const float4 lb = *reinterpret_cast<const float4*>(lp + l * 8 + 4);
      const float4 w = *reinterpret_cast<const float4*>(wp + l * 4);
      sample_point<T, VEC>(acc, vp, H, W, qid_stride, la.x, la.y, w.x);
      sample_point<T, VEC>(acc, vp, H, W, qid_stride, la.z, la.w, w.y);
      sample_point<T, VEC>(acc, vp, H, W, qid_stride, lb.x, lb.y, w.z);
      sample_point<T, VEC>(acc, vp, H, W, qid_stride, lb.z, lb.w, w.w);
    }
  } else {
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* vp = vbase + lsi[l] * (long long)qid_stride;
      for (int p = 0; p < P; ++p) {
        const T loc_w = lp[(l * P + p) * 2], loc_h = lp[(l * P + p) * 2 + 1];
        sample_point<T, VEC>(acc, vp, H, W, qid_stride, loc_w, loc_h, wp[l * P + p]);
      }
    }
  }

  T* op = out + sidx * D + cv * VEC;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    op[0] = acc[0];
  }
}

template <typename T>
int msda_dispatch(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                  const T* attw, T* out, int B, int S, int M, int D, int L, int Lq, int P,
                  hipStream_t stream) {
  OVIS_REQUIRE(value && shapes && lsi && loc && attw && out, "msda_forward: null pointer");
  OVIS_REQUIRE(B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0,
               "msda_forward: non-positive dimension (B=%d S=%d M=%d D=%d L=%d Lq=%d P=%d)", B, S, M, D,
               L, Lq, P);
  constexpr int TPB = 256;
  bool vec4 = false;
  if constexpr (sizeof(T) == 4) {
    vec4 = (D % 4 == 0) && (((uintptr_t)value | (uintptr_t)out) % 16 == 0);
  }
  const bool loc_vec = vec4 && P == 4 && (((uintptr_t)loc | (uintptr_t)attw) % 16 == 0);
  const long long n_items = (long long)B * Lq * M * (vec4 ? D / 4 : D);
  const unsigned grid = ovis::cdiv(n_items, TPB);
#define LAUNCH(VEC, LT, PT)                                                                         \
  hipLaunchKernelGGL((msda_fwd_kernel<T, VEC, LT, PT>), dim3(grid), dim3(TPB), 0, stream, value,    \
                     shapes, lsi, loc, attw, out, n_items, S, M, D, L, Lq, P)
  if constexpr (sizeof(T) == 4) {
    if (loc_vec && L == 3) LAUNCH(4, 3, 4);
    else if (loc_vec && L == 4) LAUNCH(4, 4, 4);
    else if (loc_vec && L == 1) LAUNCH(4, 1, 4);
    else if (vec4) LAUNCH(4, 0, 0);
    else LAUNCH(1, 0, 0);
  } else {
    LAUNCH(1, 0, 0);
  }
#undef LAUNCH
  return ovis::check_launch("msda_forward");
}

}  // namespace

extern "C" int ovis_msda_forward_f32(const float* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start_index, const float* sampling_loc,
                                     const float* attn_weight, float* out, int batch,
                                     int spatial_size, int num_heads, int channels, int num_levels,
                                     int num_query, int num_point, ovis_stream_t stream) {
  return msda_dispatch<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out,
                              batch, spatial_size, num_heads, channels, num_levels, num_query,
                              num_point, (hipStream_t)stream);
}

extern "C" int ovis_msda_forward_f64(const double* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start_index, const double* sampling_loc,
                                     const double* attn_weight, double* out, int batch,
                                     int spatial_size, int num_heads, int channels, int num_levels,
                                     int num_query, int num_point, ovis_stream_t stream) {
  return msda_dispatch<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                               out, batch, spatial_size, num_heads, channels, num_levels, num_query,
                               num_point, (hipStream_t)stream);
}

extern "C" int ovis_msda_forward_bf16v(const void* value_bf16, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const float* sampling_loc, const float* attn_weight, float* out, int batch, int spatial_size,
                                       int num_heads, int channels, int num_levels, int num_query, int num_point, ovis_stream_t stream) {
  return msda_dispatch16<BF16>(value_bf16, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, batch, spatial_size, num_heads,
                               channels, num_levels, num_query, num_point, (hipStream_t)stream);
}

extern "C" int ovis_msda_forward_f16v(const void* value_f16, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                      const float* sampling_loc, const float* attn_weight, float* out, int batch, int spatial_size,
                                      int num_heads, int channels, int num_levels, int num_query, int num_point, ovis_stream_t stream) {
  return msda_dispatch16<FP16>(value_f16, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, batch, spatial_size, num_heads,
                               channels, num_levels, num_query, num_point, (hipStream_t)stream);
}
