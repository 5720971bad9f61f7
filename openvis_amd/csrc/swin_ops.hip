// Swin Transformer backbone (openvis/modeling/backbone/swin.py) data movement for gfx950.
//
// The token map stays in its natural [B,H,W,C] layout; LayerNorm / qkv / proj / MLP are row-wise GEMMs on it.  Only the
// window attention needs the (padded, cyclically shifted) window grouping, so pad + roll + window_partition and
// window_reverse + roll back + crop + residual are each ONE HBM pass here (the reference materialises 4-5 copies).
// All kernels: 1 thread = 4 channels (16-byte accesses), token-major so a wavefront covers whole rows.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
window_partition_kernel(const float4* __restrict__ x, float4* __restrict__ win, int B, int H, int W, int c4n, int ws, int shift,
                        int Hp, int Wp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * Hp * Wp * c4n;
  if (i >= total) return;
  const int c = (int)(i % c4n);
  long long r = i / c4n;                       // destination token: ((b*nWy + wy)*nWx + wx)*ws*ws + iy*ws + ix
  const int ix = (int)(r % ws); r /= ws;
  const int iy = (int)(r % ws); r /= ws;
  const int nwx = Wp / ws, nwy = Hp / ws;
  const int wx = (int)(r % nwx); r /= nwx;
  const int wy = (int)(r % nwy);
  const int b = (int)(r / nwy);
  int y = wy * ws + iy + shift, xx = wx * ws + ix + shift;   // torch.roll(-shift): shifted[p] = x[(p + shift) mod n]
  if (y >= Hp) y -= Hp;
  if (xx >= Wp) xx -= Wp;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < H && xx < W) v = x[(((long long)b * H + y) * W + xx) * c4n + c];
  win[i] = v;
}

__global__ void __launch_bounds__(256)
window_merge_add_kernel(const float4* __restrict__ win, const float4* __restrict__ shortcut, float4* __restrict__ out, int B,
                        int H, int W, int c4n, int ws, int shift, int Hp, int Wp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * H * W * c4n;
  if (i >= total) return;
  const int c = (int)(i % c4n);
  long long r = i / c4n;
  const int xx = (int)(r % W); r /= W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  int ys = y - shift, xs = xx - shift;         // position in the shifted map: x[p] = shifted[(p - shift) mod n]
  if (ys < 0) ys += Hp;
  if (xs < 0) xs += Wp;
  const int nwx = Wp / ws, nwy = Hp / ws;
  const long long tok = ((((long long)b * nwy + ys / ws) * nwx + xs / ws) * ws + ys % ws) * ws + xs % ws;
  const float4 a = win[tok * c4n + c], s = shortcut[i];
  out[i] = make_float4(s.x + a.x, s.y + a.y, s.z + a.z, s.w + a.w);
}

// region id of a coordinate along one axis of the shifted map (swin.py:383-398):
//   [0, n-ws) -> 0, [n-ws, n-shift) -> 1, [n-shift, n) -> 2
__device__ __forceinline__ int region(int p, int n, int ws, int shift) { return p < n - ws ? 0 : (p < n - shift ? 1 : 2); }

__global__ void __launch_bounds__(256)
shift_mask_kernel(uint8_t* __restrict__ mask, int Hp, int Wp, int ws, int shift, int ld) {
  const int N = ws * ws;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)(Hp / ws) * (Wp / ws) * N * ld;
  if (i >= total) return;
  const int j = (int)(i % ld);
  const int q = (int)((i / ld) % N);
  const int w = (int)(i / ((long long)ld * N));
  uint8_t m = 0;
  if (j < N && shift > 0) {
    const int wy = w / (Wp / ws), wx = w % (Wp / ws);
    const int idq = region(wy * ws + q / ws, Hp, ws, shift) * 3 + region(wx * ws + q % ws, Wp, ws, shift);
    const int idj = region(wy * ws + j / ws, Hp, ws, shift) * 3 + region(wx * ws + j % ws, Wp, ws, shift);
    m = idq != idj;
  }
  mask[i] = m;
}

__global__ void __launch_bounds__(256)
patch_merge_gather_kernel(const float4* __restrict__ x, float4* __restrict__ out, int B, int H, int W, int c4n) {
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * H2 * W2 * 4 * c4n;
  if (i >= total) return;
  const int c = (int)(i % c4n);
  long long r = i / c4n;
  const int part = (int)(r % 4); r /= 4;       // 0: (2y,2x)  1: (2y+1,2x)  2: (2y,2x+1)  3: (2y+1,2x+1)
  const int xo = (int)(r % W2); r /= W2;
  const int yo = (int)(r % H2);
  const int b = (int)(r / H2);
  const int y = 2 * yo + (part & 1), xx = 2 * xo + (part >> 1);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < H && xx < W) v = x[(((long long)b * H + y) * W + xx) * c4n + c];
  out[i] = v;
}

__global__ void __launch_bounds__(256)
relpos_bias_kernel(const float* __restrict__ table, float* __restrict__ bias, int heads, int ws, int ld) {
  const int N = ws * ws;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)heads * N * ld;
  if (i >= total) return;
  const int j = (int)(i % ld);
  const int q = (int)((i / ld) % N);
  const int h = (int)(i / ((long long)ld * N));
  float v = 0.f;
  if (j < N) {
    const int idx = (q / ws - j / ws + ws - 1) * (2 * ws - 1) + (q % ws - j % ws + ws - 1);
    v = table[(long long)idx * heads + h];
  }
  bias[i] = v;
}

}  // namespace

#define SWIN_GEOM_OK(B, H, W, C, ws) ((B) > 0 && (H) > 0 && (W) > 0 && (C) > 0 && (C) % 4 == 0 && (ws) > 0)

extern "C" int ovis_swin_window_partition_f32(const float* x, float* win, int B, int H, int W, int C, int ws, int shift,
                                              ovis_stream_t stream) {
  OVIS_REQUIRE(x && win, "swin_window_partition: null pointer");
  OVIS_REQUIRE(SWIN_GEOM_OK(B, H, W, C, ws) && shift >= 0 && shift < ws, "swin_window_partition: bad geometry");
  const int Hp = ovis::cdiv(H, ws) * ws, Wp = ovis::cdiv(W, ws) * ws;
  const long long total = (long long)B * Hp * Wp * (C / 4);
  hipLaunchKernelGGL(window_partition_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(win), B, H, W, C / 4, ws, shift, Hp, Wp);
  return ovis::check_launch("swin_window_partition");
}

extern "C" int ovis_swin_window_merge_add_f32(const float* win, const float* shortcut, float* out, int B, int H, int W, int C,
                                              int ws, int shift, ovis_stream_t stream) {
  OVIS_REQUIRE(win && shortcut && out, "swin_window_merge_add: null pointer");
  OVIS_REQUIRE(SWIN_GEOM_OK(B, H, W, C, ws) && shift >= 0 && shift < ws, "swin_window_merge_add: bad geometry");
  const int Hp = ovis::cdiv(H, ws) * ws, Wp = ovis::cdiv(W, ws) * ws;
  const long long total = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(window_merge_add_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(win), reinterpret_cast<const float4*>(shortcut),
                     reinterpret_cast<float4*>(out), B, H, W, C / 4, ws, shift, Hp, Wp);
  return ovis::check_launch("swin_window_merge_add");
}

extern "C" int ovis_swin_shift_mask_u8(uint8_t* mask, int H, int W, int ws, int shift, int ld, ovis_stream_t stream) {
  OVIS_REQUIRE(mask && H > 0 && W > 0 && ws > 0 && shift >= 0 && shift < ws && ld >= ws * ws, "swin_shift_mask: bad arguments");
  const int Hp = ovis::cdiv(H, ws) * ws, Wp = ovis::cdiv(W, ws) * ws;
  const long long total = (long long)(Hp / ws) * (Wp / ws) * ws * ws * ld;
  hipLaunchKernelGGL(shift_mask_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, mask, Hp, Wp, ws, shift, ld);
  return ovis::check_launch("swin_shift_mask");
}

extern "C" int ovis_swin_patch_merge_gather_f32(const float* x, float* out, int B, int H, int W, int C, ovis_stream_t stream) {
  OVIS_REQUIRE(x && out && SWIN_GEOM_OK(B, H, W, C, 1), "swin_patch_merge_gather: bad arguments");
  const long long total = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * C;      // 4 parts x C/4 float4 per output token
  hipLaunchKernelGGL(patch_merge_gather_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), B, H, W, C / 4);
  return ovis::check_launch("swin_patch_merge_gather");
}

extern "C" int ovis_swin_relpos_bias_f32(const float* table, float* bias, int heads, int ws, int ld, ovis_stream_t stream) {
  OVIS_REQUIRE(table && bias && heads > 0 && ws > 0 && ld >= ws * ws, "swin_relpos_bias: bad arguments");
  const long long total = (long long)heads * ws * ws * ld;
  hipLaunchKernelGGL(relpos_bias_kernel, dim3(ovis::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, table, bias, heads, ws, ld);
  return ovis::check_launch("swin_relpos_bias");
}
