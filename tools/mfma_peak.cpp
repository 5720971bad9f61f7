// What the matrix cores of this MI355X deliver on RANDOM fp16 operands with no memory traffic at all: a bare loop of
// v_mfma_f32_16x16x32_f16 on register operands, 16 accumulators per wavefront, 1 or 2 wavefronts per SIMD, every CU busy.
// The chip lowers its clock under MFMA load (MI355X_MICROARCH.md, DVFS give-back), so this -- not 2.5 PF -- is the ceiling a
// GEMM kernel on real activations can approach; the same loop on zero operands shows the clock it would otherwise hold.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_peak.cpp -o build/mfma_peak && ./build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__global__ void __launch_bounds__(512) mfma_loop(const f16x8* __restrict__ src, float* __restrict__ sink, int iters,
                                                 unsigned long long* __restrict__ clk) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(i * 64 + lane) & 1023]; b[i] = src[((i + 4) * 64 + lane) & 1023]; }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = s[0];       // keeps the accumulators live
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = r1 - r0; clk[blockIdx.x * 2 + 1] = c1 - c0; }
}

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  f16x8* src; float* sink; unsigned long long* clk;
  OK(hipMalloc(&src, 1024 * sizeof(f16x8))); OK(hipMalloc(&sink, 4)); OK(hipMalloc(&clk, 256 * 16));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero) {
    std::vector<_Float16> h(8192);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = zero ? (_Float16)0.f : (_Float16)(((x >> 8) * (1.f / 8388608.f) - 1.f)); }
    OK(hipMemcpy(src, h.data(), 8192 * 2, hipMemcpyHostToDevice));
    for (int waves = 4; waves <= 8; waves += 4) {
      const int iters = 20000;                                        // 16 MFMAs each: ~2.5-5 ms per launch
      double best = 0, ghz = 0;
      for (int r = 0; r < 60; ++r) {                                  // ~0.2-0.3 s of back-to-back launches before the last (reported) ones
        OK(hipEventRecord(e0, 0));
        mfma_loop<<<256, waves * 64, 0, 0>>>(src, sink, iters, clk);
        OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 50) {
          const double tf = 256.0 * waves * iters * 16 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
          std::vector<unsigned long long> c(512); OK(hipMemcpy(c.data(), clk, 512 * 8, hipMemcpyDeviceToHost));
          std::vector<double> g; for (int b = 0; b < 256; ++b) g.push_back((double)c[2 * b + 1] / (double)c[2 * b] * 0.1);
          std::sort(g.begin(), g.end());
          if (tf > best) { best = tf; ghz = g[128]; }
        }
      }
      printf("mfma_f32_16x16x32_f16 bare loop, %s operands, %d wavefront(s) per SIMD: %.0f TFLOP/s, in-kernel clock %.3f GHz\n",
             zero ? "zero" : "random", waves / 4, best, ghz);
    }
  }
  return 0;
}
