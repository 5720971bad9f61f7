// A-operand loaders shared by the f32 and the f32->fp16-converting GEMM kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace ovis {

struct ConvGeom {
  int H, W, Cin, OH, OW, KH, KW, stride, pad;
};

// ---- A-operand loaders: fetch 4 consecutive k of row m as a float4 (zero outside) ----------------
// Loads are BRANCH-FREE (clamped address + select): a predicated load compiles to an exec-masked branch with its
// own s_waitcnt, which serialises the 8 staging loads of a K tile (2x slower GEMM when measured).
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// dense row-major [M,K]; VEC: K % 4 == 0, lda % 4 == 0 and 16-byte aligned base (a chunk is all-in or all-out)
template <bool VEC>
struct DenseA {
  const float* A;
  long long lda;
  int M, K;
  __device__ __forceinline__ void advance(long long elems) { A += elems; }   // batched GEMM: blockIdx.y * a_bs
  // hoisted form (the kernels call row() once per staged row before the K loop and kctx() once per K step)
  struct RowCtx { int m; };
  struct KCtx { int k; };
  __device__ __forceinline__ RowCtx row(int m) const { return RowCtx{m}; }
  __device__ __forceinline__ KCtx kctx(int k) const { return KCtx{k}; }
  __device__ __forceinline__ float4 load(const RowCtx& r, const KCtx& kc, bool& ok) const { return load(r.m, kc.k, ok); }
  // raw load from a clamped address; `ok` tells the caller whether to keep it (selected at LDS-store time so the
  // s_waitcnt for this load lands AFTER the current tile's MFMAs, not right behind the load)
  __device__ __forceinline__ float4 load(int m, int k, bool& ok) const {
    if constexpr (VEC) {
      ok = m < M && k < K;
      const float* p = A + (ok ? (long long)m * lda + k : 0);
      return *reinterpret_cast<const float4*>(p);
    } else {
      ok = true;
      const bool okm = m < M;
      const float* row = A + (okm ? (long long)m * lda : 0);
      float4 v;
      v.x = (okm && k < K) ? row[min(k, K - 1)] : 0.f;
      v.y = (okm && k + 1 < K) ? row[min(k + 1, K - 1)] : 0.f;
      v.z = (okm && k + 2 < K) ? row[min(k + 2, K - 1)] : 0.f;
      v.w = (okm && k + 3 < K) ? row[min(k + 3, K - 1)] : 0.f;
      return v;
    }
  }
};

// dense row-major fp16 [M,K] (K % 8 == 0, lda % 8 == 0, 16-byte aligned base): 8 consecutive k of a row are ONE 16-byte load, already in
// the MFMA's operand format (activations stored in fp16 between the convolutions of a bottleneck: conv_h16.hip, gemm_f16cvt.hip)
struct DenseH {
  const _Float16* A;
  long long lda;
  int M, K;
  __device__ __forceinline__ void advance(long long elems) { A += elems; }
  struct RowCtx { int m; };
  __device__ __forceinline__ RowCtx row(int m) const { return RowCtx{m}; }
  __device__ __forceinline__ uint4 load8(const RowCtx& r, int k, bool& ok) const {
    ok = r.m < M && k < K;
    return *reinterpret_cast<const uint4*>(A + (ok ? (long long)r.m * lda + k : 0));
  }
};

// Two A sources along K (round 4): columns [0, K1) from a dense fp16 matrix [M, K1], columns [K1, K) from a second map whose row m starts at
// element base2(m) -- a dense fp16 matrix (A2H) or the f32 pixels (t, s oy, s ox) of an NHWC map [T, H, W, C2] (a 1x1 / stride-s convolution's
// rows).  conv3 and the projection shortcut of a ResNet bottleneck (detectron2 BottleneckBlock.forward: out = conv3(out) + shortcut(x)) are
// then ONE GEMM over the concatenated K axis: the shortcut tensor is neither written nor read back.  K1 % 64 == 0: a 64-wide K tile lies in
// one source, so the choice is uniform per tile.
template <bool A2H>
struct DualA {
  const _Float16* A1; long long lda1; int K1;
  const void* A2; long long lda2;            // A2H: dense fp16 rows (lda2 elements apart); else the f32 map X
  int H, W, C2, OH, OW, stride;              // !A2H: geometry of the strided pixel rows
  int M, K;
  __device__ __forceinline__ void advance(long long) {}
  struct RowCtx { int m; long long base2; };
  __device__ __forceinline__ RowCtx row(int m) const {
    RowCtx r; r.m = m;
    if constexpr (A2H) r.base2 = (long long)m * lda2;
    else { const int ox = m % OW, t = m / OW, oy = t % OH, n = t / OH; r.base2 = (((long long)n * H + oy * stride) * W + ox * stride) * C2; }
    return r;
  }
  // first source (and a fp16 second one): 8 halfs of row r at column k as raw bits
  __device__ __forceinline__ uint4 load8(const RowCtx& r, int k, bool& ok) const {
    ok = r.m < M && k < K;
    const bool first = k < K1;
    const _Float16* p = first ? A1 + (ok ? (long long)r.m * lda1 + k : 0) : reinterpret_cast<const _Float16*>(A2) + (ok ? r.base2 + (k - K1) : 0);
    return *reinterpret_cast<const uint4*>(p);
  }
  // f32 second source: 4 floats of row r at column k (k >= K1)
  __device__ __forceinline__ float4 load4(const RowCtx& r, int k, bool& ok) const {
    ok = r.m < M && k < K;
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(A2) + (ok ? r.base2 + (k - K1) : 0));
  }
};

// Implicit im2col over an NHWC input; k = (kh*KW + kw)*Cin + c, Cin % 4 == 0 (float4 never straddles a pixel).
struct ConvA {
  const float* X;
  ConvGeom g;
  int M, K;
  __device__ __forceinline__ void advance(long long) {}
  // Hoisted form: load(m, k) below costs five runtime integer divisions per 16-byte load (m -> n, oh, ow; k -> kh, kw, c) --
  // 40 per thread and K step in the 128x128x64 kernels, several times the tile's MFMA time.  A staged row's pixel is fixed for the
  // whole K loop and the 8 consecutive k of a thread are one tap (Cin % 4 == 0: a float4 never straddles a pixel), so the kernels
  // decompose each row ONCE (row()) and each K step's two k offsets once (kctx()); a load is then adds and compares.
  struct RowCtx { long long base; int ih0, iw0; bool okm; };        // element offset of input pixel (n, oh*stride - pad, ow*stride - pad)
  struct KCtx { int off, kh, kw; bool okk; };                        // (kh*W + kw)*Cin + c
  __device__ __forceinline__ RowCtx row(int m) const {
    const int ow = m % g.OW;
    const int t = m / g.OW;
    const int oh = t % g.OH;
    const int n = t / g.OH;
    RowCtx r;
    r.ih0 = oh * g.stride - g.pad; r.iw0 = ow * g.stride - g.pad; r.okm = m < M;
    r.base = (((long long)n * g.H + r.ih0) * g.W + r.iw0) * g.Cin;
    return r;
  }
  __device__ __forceinline__ KCtx kctx(int k) const {
    const int c = k % g.Cin;
    const int t2 = k / g.Cin;
    KCtx kc;
    kc.kw = t2 % g.KW; kc.kh = t2 / g.KW; kc.okk = k < K;
    kc.off = (kc.kh * g.W + kc.kw) * g.Cin + c;
    return kc;
  }
  __device__ __forceinline__ float4 load(const RowCtx& r, const KCtx& kc, bool& ok) const {
    const int ih = r.ih0 + kc.kh, iw = r.iw0 + kc.kw;
    ok = r.okm && kc.okk && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
    const float* p = X + (ok ? r.base + kc.off : 0);
    return *reinterpret_cast<const float4*>(p);
  }
  __device__ __forceinline__ float4 load(int m, int k, bool& ok) const {
    const int ow = m % g.OW;
    const int t = m / g.OW;
    const int oh = t % g.OH;
    const int n = t / g.OH;
    const int c = k % g.Cin;
    const int t2 = k / g.Cin;
    const int kw = t2 % g.KW;
    const int kh = t2 / g.KW;
    const int ih = oh * g.stride - g.pad + kh;
    const int iw = ow * g.stride - g.pad + kw;
    ok = m < M && k < K && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
    const float* p = X + (ok ? (((long long)n * g.H + ih) * g.W + iw) * g.Cin + c : 0);
    return *reinterpret_cast<const float4*>(p);
  }
};

}  // namespace ovis
