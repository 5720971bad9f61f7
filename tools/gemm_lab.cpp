// gemm_lab: standalone (no torch) correctness + timing harness for ovis_gemm_nt_f16 on the CLIP ViT shapes.
//   build:  make -C openvis_amd/csrc lab      ->  build/gemm_lab
//   run  :  build/gemm_lab [check] [time] [variants...]
// Variants are (mode, raster_group, desync_ns) triples of ovis_set_f16_gemm_mode; timing is interleaved rounds of all
// variants in one process on uniform random [-1,1) operands (cdna_hip_programming.md rules 24/25).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/openvis_hip.h"
extern "C" int ovis_set_f32a_pp(int on);
extern "C" int ovis_pp_tile_rows(int tm);
extern "C" const char* ovis_gemm_nt_f32_w3_kernel(const float* A, long long lda, const void* W3, long long ldb, long long plane, const float* C, long long ldc, int M, int N, int K, const float* bias, const float* residual, long long ldr, int act);
extern "C" int ovis_pp_debug(int flags, unsigned long long* stamps);   // lab-only entry of gemm_f16_pp.hip
extern "C" int ovis_pp_epilogue(int epi);                               // epilogue variant (variant field 4, bits 8..)

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define OVIS_OKAY(x) do { int r_ = (x); if (r_ != 0) { printf("ovis error %d: %s at %s:%d\n", r_, ovis_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// kind 0: uniform [-scale, scale); kind 1: integers in {-1, 0, 1}
__global__ void fill_f16(_Float16* p, long long n, unsigned seed, int kind, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned h = hash32((unsigned)i * 2654435761U + seed);
    p[i] = kind == 1 ? (_Float16)(float)((int)(h % 3u) - 1) : (_Float16)(((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
  }
}
__global__ void fill_f32(float* p, long long n, unsigned seed, int kind, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned h = hash32((unsigned)i * 2246822519U + seed);
    p[i] = kind == 1 ? (float)((int)(h % 5u) - 2) : ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
  }
}
// naive reference: one thread per output, f32 fmaf chain over k (exact on the integer data)
__global__ void ref_gemm(const _Float16* A, const _Float16* B, float* C, int M, int N, int K, const float* bias, const float* R, int act) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = (bias ? bias[n] : 0.f) + (R ? R[i] : 0.f);
  for (int k = 0; k < K; ++k) s = fmaf((float)A[(long long)m * K + k], (float)B[(long long)n * K + k], s);
  if (act == 1) s = fmaxf(s, 0.f);
  else if (act == 2) s = s / (1.f + expf(-1.702f * s));
  C[i] = s;
}
__global__ void diff_kernel(const float* ref, const void* out, int out_f16, long long n, float* maxabs, unsigned long long* nbad, float tol_abs, float tol_rel) {
  float mx = 0.f; unsigned long long bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float o = out_f16 ? (float)((const _Float16*)out)[i] : ((const float*)out)[i];
    const float d = fabsf(o - ref[i]);
    if (!(d <= tol_abs + tol_rel * fabsf(ref[i]))) ++bad;
    mx = fmaxf(mx, d == d ? d : 1e30f);
  }
  atomicMax((int*)maxabs, __float_as_int(mx));
  if (bad) atomicAdd(nbad, bad);
}

__global__ void bitdiff_kernel(const unsigned short* a, const unsigned short* b, long long n, unsigned long long* nbad) {
  unsigned long long bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) bad += a[i] != b[i];
  if (bad) atomicAdd(nbad, bad);
}

// ---- x3 (f32-grade GEMM on bf16 planes) -------------------------------------------------------------------------------
__global__ void ref_gemm_f64(const float* A, const float* B, double* C, double* S, int M, int N, int K, const float* bias, const float* R, int act) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  double s = (bias ? (double)bias[n] : 0.0) + (R ? (double)R[i] : 0.0), sc = fabs(s);
  for (int k = 0; k < K; ++k) { const double a = A[(long long)m * K + k], b = B[(long long)n * K + k]; s += a * b; sc += fabs(a * b); }
  if (act == 1) s = s > 0 ? s : 0;
  C[i] = s; S[i] = sc;
}
__global__ void x3_err_kernel(const double* ref, const double* scale, const float* out, const __bf16* planes, long long plane, long long n, float* maxerr) {
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double o = out ? (double)out[i] : (double)(float)planes[i] + (double)(float)planes[plane + i] + (double)(float)planes[2 * plane + i];
    const double e = fabs(o - ref[i]) / scale[i];
    mx = fmaxf(mx, e == e ? (float)e : 1e30f);
  }
  atomicMax((int*)maxerr, __float_as_int(mx));
}

static int run_x3(hipStream_t s, bool do_time) {
  int fails = 0;
  float* d_max; HIP_OK(hipMalloc(&d_max, 4));
  struct Case { int M, N, K, act, res, out_planes; const char* name; };
  const Case cases[] = {{66000 + 37, 256, 256, 0, 1, 0, "value/out proj"}, {66000 + 37, 1024, 256, 1, 0, 0, "FFN1 f32 out"},
                        {66000 + 37, 1024, 256, 1, 0, 1, "FFN1 planes out"}, {66000 + 37, 256, 1024, 0, 1, 0, "FFN2"}, {33000, 520, 320, 0, 0, 0, "ragged"}};
  for (const Case& c : cases) {
    const long long MN = (long long)c.M * c.N, MK = (long long)c.M * c.K, NK = (long long)c.N * c.K;
    float *A, *W, *bias, *R = nullptr, *C; double *ref, *sc; __bf16 *A3, *W3, *C3 = nullptr;
    HIP_OK(hipMalloc(&A, MK * 4)); HIP_OK(hipMalloc(&W, NK * 4)); HIP_OK(hipMalloc(&bias, c.N * 4)); HIP_OK(hipMalloc(&C, MN * 4));
    HIP_OK(hipMalloc(&ref, MN * 8)); HIP_OK(hipMalloc(&sc, MN * 8)); HIP_OK(hipMalloc(&A3, MK * 6)); HIP_OK(hipMalloc(&W3, NK * 6));
    fill_f32<<<2048, 256, 0, s>>>(A, MK, 3u, 0, 4.f); fill_f32<<<512, 256, 0, s>>>(W, NK, 9u, 0, 1.f / sqrtf((float)c.K)); fill_f32<<<8, 256, 0, s>>>(bias, c.N, 5u, 0, 1.f);
    if (c.res) { HIP_OK(hipMalloc(&R, MN * 4)); fill_f32<<<2048, 256, 0, s>>>(R, MN, 7u, 0, 2.f); }
    if (c.out_planes) HIP_OK(hipMalloc(&C3, MN * 6));
    ref_gemm_f64<<<(unsigned)((MN + 255) / 256), 256, 0, s>>>(A, W, ref, sc, c.M, c.N, c.K, bias, R, c.act);
    OVIS_OKAY(ovis_split_f32_to_bf16x3_v8(A, A3, MK, s)); OVIS_OKAY(ovis_split_f32_to_bf16x3_v8(W, W3, NK, s));
    for (int rep = 0; rep < 2; ++rep) {
      HIP_OK(hipMemsetAsync(c.out_planes ? (void*)C3 : (void*)C, 0xff, c.out_planes ? MN * 6 : MN * 4, s));
      OVIS_OKAY(ovis_gemm_nt_bf16x3_planes(A3, c.K, MK, W3, c.K, NK, c.out_planes ? (void*)C3 : (void*)C, c.N, MN, c.M, c.N, c.K, bias, R, c.N, c.act, c.out_planes, s));
      HIP_OK(hipMemsetAsync(d_max, 0, 4, s));
      x3_err_kernel<<<1024, 256, 0, s>>>(ref, sc, c.out_planes ? nullptr : C, C3, MN, MN, d_max);
      float mx; HIP_OK(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
      const bool ok = mx < 2e-6f;
      if (!ok) ++fails;
      if (rep == 0 || !ok) printf("x3 check %-16s M=%d N=%d K=%d act=%d res=%d planes=%d: max err / sum|ab| = %.3g %s\n", c.name, c.M, c.N, c.K, c.act, c.res, c.out_planes, mx, ok ? "OK" : "FAIL");
    }
    // the shipped bf16x3 kernel (pre-split weights, activation split in the loop) on the same data
    OVIS_OKAY(ovis_gemm_nt_f32_w3(A, c.K, W, c.K, W3, NK, C, c.N, c.M, c.N, c.K, bias, R, c.N, c.act, s));
    HIP_OK(hipMemsetAsync(d_max, 0, 4, s));
    x3_err_kernel<<<1024, 256, 0, s>>>(ref, sc, C, nullptr, 0, MN, d_max);
    float mx; HIP_OK(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
    printf("x3 check %-16s legacy gemm_f32x3_kernel: max err / sum|ab| = %.3g\n", c.name, mx);
    if (!c.out_planes) {   // bf16x2 (three products): gemm_f32x3_kernel<.., 2> against the ping-pong kernel's f32-A mode, three runs each
      OVIS_OKAY(ovis_set_f32_gemm_mode(2));
      for (int pp = 0; pp < 2; ++pp) {
        OVIS_OKAY(ovis_set_f32a_pp(pp));
        const char* kn = ovis_gemm_nt_f32_w3_kernel(A, c.K, W3, c.K, NK, C, c.N, c.M, c.N, c.K, bias, R, c.N, c.act);
        for (int rep = 0; rep < 3; ++rep) {
          HIP_OK(hipMemsetAsync(C, 0xff, MN * 4, s));
          OVIS_OKAY(ovis_gemm_nt_f32_w3(A, c.K, W, c.K, W3, NK, C, c.N, c.M, c.N, c.K, bias, R, c.N, c.act, s));
          HIP_OK(hipMemsetAsync(d_max, 0, 4, s));
          x3_err_kernel<<<1024, 256, 0, s>>>(ref, sc, C, nullptr, 0, MN, d_max);
          float m2; HIP_OK(hipMemcpyAsync(&m2, d_max, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
          const bool ok = m2 < 3.1e-5f;                              // 2^-15: two operands of 16 significand bits
          if (!ok) ++fails;
          if (rep == 0 || !ok) printf("x2 check %-16s %s [%s]: max err / sum|ab| = %.3g %s\n", c.name, pp ? "ping-pong f32-A" : "gemm_f32x3_kernel<..,2>", kn, m2, ok ? "OK" : "FAIL");
        }
      }
      OVIS_OKAY(ovis_set_f32a_pp(1)); OVIS_OKAY(ovis_set_f32_gemm_mode(1));
    }
    HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(bias)); HIP_OK(hipFree(C)); HIP_OK(hipFree(ref)); HIP_OK(hipFree(sc)); HIP_OK(hipFree(A3)); HIP_OK(hipFree(W3));
    if (R) HIP_OK(hipFree(R));
    if (C3) HIP_OK(hipFree(C3));
  }
  if (do_time) {
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    struct Shape { int M, N, K, act, res; const char* name; };
    const Shape shapes[] = {{96600, 256, 256, 0, 1, "proj256"}, {96600, 288, 256, 0, 0, "offs288"}, {96600, 544, 256, 0, 0, "fused544"}, {96600, 1024, 256, 1, 0, "ffn1"},
                            {96600, 256, 1024, 0, 1, "ffn2"}, {294400, 256, 256, 0, 0, "maskfeat"}, {96600, 256, 2048, 0, 0, "inproj2048"}};
    for (const Shape& sh : shapes) {
      const long long MN = (long long)sh.M * sh.N, MK = (long long)sh.M * sh.K, NK = (long long)sh.N * sh.K;
      float *A, *W, *bias, *R = nullptr, *C; __bf16 *A3, *W3, *C3;
      HIP_OK(hipMalloc(&A, MK * 4)); HIP_OK(hipMalloc(&W, NK * 4)); HIP_OK(hipMalloc(&bias, sh.N * 4)); HIP_OK(hipMalloc(&C, MN * 4));
      HIP_OK(hipMalloc(&A3, MK * 6)); HIP_OK(hipMalloc(&W3, NK * 6)); HIP_OK(hipMalloc(&C3, MN * 6));
      fill_f32<<<2048, 256, 0, s>>>(A, MK, 3u, 0, 4.f); fill_f32<<<512, 256, 0, s>>>(W, NK, 9u, 0, 1.f / sqrtf((float)sh.K)); fill_f32<<<8, 256, 0, s>>>(bias, sh.N, 5u, 0, 1.f);
      if (sh.res) { HIP_OK(hipMalloc(&R, MN * 4)); fill_f32<<<2048, 256, 0, s>>>(R, MN, 7u, 0, 2.f); }
      OVIS_OKAY(ovis_split_f32_to_bf16x3_v8(W, W3, NK, s));
      const bool elig = ovis_gemm_x3pp_eligible(sh.M, sh.N, sh.K, 1) != 0;
      double t[4] = {1e30, 1e30, 1e30, 1e30};
      for (int r = 0; r < 4; ++r)
        for (int v = 0; v < 4; ++v) {
          if (v > 0 && !elig) continue;
          HIP_OK(hipEventRecord(e0, s));
          for (int it = 0; it < 10; ++it) {
            if (v == 0) OVIS_OKAY(ovis_gemm_nt_f32_w3(A, sh.K, W, sh.K, W3, NK, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, sh.act, s));
            else if (v == 1) OVIS_OKAY(ovis_split_f32_to_bf16x3_v8(A, A3, MK, s));
            else if (v == 2) OVIS_OKAY(ovis_gemm_nt_bf16x3_planes(A3, sh.K, MK, W3, sh.K, NK, C, sh.N, MN, sh.M, sh.N, sh.K, bias, R, sh.N, sh.act, 0, s));
            else if (!sh.res) OVIS_OKAY(ovis_gemm_nt_bf16x3_planes(A3, sh.K, MK, W3, sh.K, NK, C3, sh.N, MN, sh.M, sh.N, sh.K, bias, nullptr, sh.N, sh.act, 1, s));
          }
          HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
          float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0) t[v] = std::min(t[v], (double)ms / 10);
        }
      double t2[2] = {1e30, 1e30};                                  // bf16x2: gemm_f32x3_kernel<..,2> | ping-pong f32-A
      OVIS_OKAY(ovis_set_f32_gemm_mode(2));
      for (int r = 0; r < 4; ++r)
        for (int pp = 0; pp < 2; ++pp) {
          OVIS_OKAY(ovis_set_f32a_pp(pp));
          HIP_OK(hipEventRecord(e0, s));
          for (int it = 0; it < 10; ++it) OVIS_OKAY(ovis_gemm_nt_f32_w3(A, sh.K, W, sh.K, W3, NK, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, sh.act, s));
          HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
          float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0) t2[pp] = std::min(t2[pp], (double)ms / 10);
        }
      OVIS_OKAY(ovis_set_f32a_pp(1)); OVIS_OKAY(ovis_set_f32_gemm_mode(1));
      printf("x2 time %-10s M=%d N=%d K=%d: gemm_f32x3_kernel<..,2> %.4f ms (%.0f TF f32-eq) | ping-pong f32-A %.4f ms (%.0f TF) [%s]\n", sh.name, sh.M, sh.N, sh.K,
             t2[0], 2.0 * sh.M * sh.N * sh.K / t2[0] / 1e9, t2[1], 2.0 * sh.M * sh.N * sh.K / t2[1] / 1e9,
             ovis_gemm_nt_f32_w3_kernel(A, sh.K, W3, sh.K, NK, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, sh.act));
      const double fl = 2.0 * sh.M * sh.N * sh.K;
      printf("x3 time %-10s M=%d N=%d K=%d: legacy %.4f ms (%.0f TF f32-eq) | split A %.4f ms | pp f32-out %.4f ms (%.0f TF) | pp planes-out %.4f ms\n", sh.name,
             sh.M, sh.N, sh.K, t[0], fl / t[0] / 1e9, t[1], t[2], fl / t[2] / 1e9, sh.res ? 0.0 : t[3]);
      HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(bias)); HIP_OK(hipFree(C)); HIP_OK(hipFree(A3)); HIP_OK(hipFree(W3)); HIP_OK(hipFree(C3));
      if (R) HIP_OK(hipFree(R));
    }
  }
  HIP_OK(hipFree(d_max));
  return fails;
}

struct Variant { int mode, grp, desync, dbg; std::string name; };

int main(int argc, char** argv) {
  bool do_check = false, do_time = false, do_trace = false, do_x3 = false, do_r16 = false;
  float time_scale = 1.f;   // `zeros`: zero-filled operands in the timing runs (data-dependent power -> clocks)
  std::vector<Variant> variants;
  int iters = 10, rounds = 3;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "check")) do_check = true;
    else if (!strcmp(argv[i], "time")) do_time = true;
    else if (!strncmp(argv[i], "iters=", 6)) iters = atoi(argv[i] + 6);
    else if (!strncmp(argv[i], "rounds=", 7)) rounds = atoi(argv[i] + 7);
    else if (!strcmp(argv[i], "trace")) do_trace = true;
    else if (!strcmp(argv[i], "x3")) do_x3 = true;
    else if (!strcmp(argv[i], "r16")) do_r16 = true;
    else if (!strcmp(argv[i], "zeros")) time_scale = 0.f;
    else { Variant v; v.dbg = 0; if (sscanf(argv[i], "%d,%d,%d,%d", &v.mode, &v.grp, &v.desync, &v.dbg) >= 3) { v.name = argv[i]; variants.push_back(v); } }
  }
  if (variants.empty()) { variants.push_back({0, 0, 0, 0, "0,0,0"}); variants.push_back({1, 6, 0, 0, "1,6,0"}); }
  hipStream_t s; HIP_OK(hipStreamCreate(&s));
  float* d_max; unsigned long long* d_bad;
  HIP_OK(hipMalloc(&d_max, 4)); HIP_OK(hipMalloc(&d_bad, 8));
  int fails = 0;
  if (do_x3) fails += run_x3(s, true);
  if (do_r16) {   // out-proj / c_proj on the fp16 residual stream: tile height 256 vs 192 vs automatic, interleaved
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    struct Shape { int M, N, K; const char* name; };
    const Shape shapes[] = {{94600, 768, 768, "outproj16"}, {94600, 768, 3072, "fc2_16"}, {98500, 768, 768, "outproj16b"}, {98500, 768, 3072, "fc2_16b"}};
    for (const Shape& sh : shapes) {
      _Float16 *A, *B, *R, *C; float* bias;
      const long long MN = (long long)sh.M * sh.N;
      HIP_OK(hipMalloc(&A, (size_t)sh.M * sh.K * 2)); HIP_OK(hipMalloc(&B, (size_t)sh.N * sh.K * 2)); HIP_OK(hipMalloc(&R, MN * 2));
      HIP_OK(hipMalloc(&C, MN * 2)); HIP_OK(hipMalloc(&bias, sh.N * 4));
      fill_f16<<<2048, 256, 0, s>>>(A, (long long)sh.M * sh.K, 11u, 0, 1.f); fill_f16<<<2048, 256, 0, s>>>(B, (long long)sh.N * sh.K, 23u, 0, 1.f / sqrtf((float)sh.K));
      fill_f16<<<2048, 256, 0, s>>>(R, MN, 7u, 0, 1.f); fill_f32<<<64, 256, 0, s>>>(bias, sh.N, 5u, 0, 1.f);
      // epilogue variants (EPI): bit-identical outputs expected (same arithmetic, different store / residual-load shapes), then interleaved timing
      const int epis[4] = {0, 1, 4, 5};
      _Float16* C0; HIP_OK(hipMalloc(&C0, MN * 2));
      for (int tmv = 0; tmv < 2; ++tmv) {
        ovis_pp_tile_rows(tmv ? 192 : 256);
        ovis_pp_epilogue(0);
        HIP_OK(hipMemsetAsync(C0, 0xff, MN * 2, s));
        OVIS_OKAY(ovis_gemm_nt_f16_res16(A, sh.K, B, sh.K, C0, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, s));
        for (int v = 1; v < 4; ++v) {
          ovis_pp_epilogue(epis[v]);
          HIP_OK(hipMemsetAsync(C, 0xee, MN * 2, s));
          OVIS_OKAY(ovis_gemm_nt_f16_res16(A, sh.K, B, sh.K, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, s));
          HIP_OK(hipMemsetAsync(d_bad, 0, 8, s));
          bitdiff_kernel<<<1024, 256, 0, s>>>(reinterpret_cast<const unsigned short*>(C0), reinterpret_cast<const unsigned short*>(C), MN, d_bad);
          unsigned long long bad; HIP_OK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
          if (bad) ++fails;
          printf("r16 check %-10s TM %d EPI %d vs EPI 0: %llu differing elements %s\n", sh.name, tmv ? 192 : 256, epis[v], bad, bad ? "FAIL" : "OK");
        }
      }
      HIP_OK(hipFree(C0));
      ovis_pp_tile_rows(0);
      double best[4] = {1e30, 1e30, 1e30, 1e30};
      std::vector<double> all[4];
      for (int r = 0; r < rounds + 1; ++r)
        for (int v = 0; v < 4; ++v) {
          ovis_pp_epilogue(epis[v]);
          HIP_OK(hipEventRecord(e0, s));
          for (int it = 0; it < iters; ++it) OVIS_OKAY(ovis_gemm_nt_f16_res16(A, sh.K, B, sh.K, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, s));
          HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
          float t; HIP_OK(hipEventElapsedTime(&t, e0, e1));
          if (r > 0) { best[v] = std::min(best[v], (double)t / iters); all[v].push_back((double)t / iters); }
        }
      ovis_pp_epilogue(0);
      const double fl = 2.0 * sh.M * sh.N * sh.K;
      for (int v = 0; v < 4; ++v) {
        std::sort(all[v].begin(), all[v].end());
        const double med = all[v][all[v].size() / 2];
        printf("r16 time %-10s M=%d N=%d K=%d EPI %d (auto TM): median %.4f ms (%.0f TF)  min %.4f ms (%.0f TF)\n", sh.name, sh.M, sh.N, sh.K, epis[v], med, fl / med / 1e9, best[v], fl / best[v] / 1e9);
      }
      HIP_OK(hipFree(A)); HIP_OK(hipFree(B)); HIP_OK(hipFree(R)); HIP_OK(hipFree(C)); HIP_OK(hipFree(bias));
    }
  }

  if (do_check) {
    struct Case { int M, N, K, act, out16, bias, res, kind; };
    const Case cases[] = {
      {8192 + 37, 2304, 768, 0, 1, 1, 0, 1},      // QKV shape, ragged M, exact integers
      {8192 + 37, 2304, 768, 0, 1, 1, 0, 0},      // random floats
      {7000, 3072, 768, 2, 1, 1, 0, 0},           // fc1 + QuickGELU
      {22000, 768, 3072, 0, 0, 1, 1, 1},          // fc2: f32 out, residual, exact integers
      {22000, 768, 768, 0, 0, 1, 1, 0},           // out-proj: f32 out, residual
      {9000, 2056, 832, 0, 0, 0, 0, 0},           // ragged N (8-multiple), odd K steps (13), no bias
      {9000, 2056, 832, 1, 1, 1, 0, 0},           // ReLU, fp16 out, ragged N
    };
    for (const Case& c : cases) {
      _Float16 *A, *B; float *bias = nullptr, *R = nullptr, *ref; void* C;
      const long long MN = (long long)c.M * c.N;
      HIP_OK(hipMalloc(&A, (size_t)c.M * c.K * 2)); HIP_OK(hipMalloc(&B, (size_t)c.N * c.K * 2));
      HIP_OK(hipMalloc(&ref, MN * 4)); HIP_OK(hipMalloc(&C, MN * 4));
      fill_f16<<<2048, 256, 0, s>>>(A, (long long)c.M * c.K, 11u, c.kind, 1.f);
      fill_f16<<<2048, 256, 0, s>>>(B, (long long)c.N * c.K, 23u, c.kind, c.kind ? 1.f : 1.f / sqrtf((float)c.K));
      if (c.bias) { HIP_OK(hipMalloc(&bias, c.N * 4)); fill_f32<<<64, 256, 0, s>>>(bias, c.N, 5u, c.kind, 1.f); }
      if (c.res) { HIP_OK(hipMalloc(&R, MN * 4)); fill_f32<<<2048, 256, 0, s>>>(R, MN, 7u, c.kind, 1.f); }
      ref_gemm<<<(unsigned)((MN + 255) / 256), 256, 0, s>>>(A, B, ref, c.M, c.N, c.K, bias, R, c.act);
      for (const Variant& v : variants) {
        OVIS_OKAY(ovis_set_f16_gemm_mode(v.mode, v.grp, v.desync));
        if (v.dbg & 38) { printf("dbg flags 2 / 4 / 32 were removed from the kernel (results: profiles/r03/lab_l2_locality.txt, lab_skeleton.txt)\n"); continue; }
        ovis_pp_debug(v.dbg & 255, nullptr); ovis_pp_epilogue(v.dbg >> 8);
        for (int rep = 0; rep < 3; ++rep) {                       // repeated: a race shows up as run-to-run differences
          HIP_OK(hipMemsetAsync(C, 0xff, MN * 4, s));
          OVIS_OKAY(ovis_gemm_nt_f16(A, c.K, B, c.K, C, c.N, c.M, c.N, c.K, bias, R, c.N, c.act, c.out16, s));
          HIP_OK(hipMemsetAsync(d_max, 0, 4, s)); HIP_OK(hipMemsetAsync(d_bad, 0, 8, s));
          const float tol_abs = c.kind ? 0.f : (c.out16 ? 2e-2f : 2e-3f), tol_rel = c.kind ? 0.f : (c.out16 ? 2e-3f : 1e-4f);
          diff_kernel<<<1024, 256, 0, s>>>(ref, C, c.out16, MN, d_max, d_bad, tol_abs, tol_rel);
          float mx; unsigned long long bad;
          HIP_OK(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
          HIP_OK(hipStreamSynchronize(s));
          const bool ok = bad == 0;
          if (!ok) ++fails;
          if (rep == 0 || !ok)
            printf("check M=%d N=%d K=%d act=%d out16=%d bias=%d res=%d %s variant=%s rep=%d: max|d|=%.3g bad=%llu %s\n", c.M, c.N, c.K, c.act,
                   c.out16, c.bias, c.res, c.kind ? "int" : "rnd", v.name.c_str(), rep, mx, bad, ok ? "OK" : "FAIL");
        }
      }
      HIP_OK(hipFree(A)); HIP_OK(hipFree(B)); HIP_OK(hipFree(ref)); HIP_OK(hipFree(C));
      if (bias) HIP_OK(hipFree(bias));
      if (R) HIP_OK(hipFree(R));
    }
  }

  if (do_time) {
    struct Shape { int M, N, K, act, out16, res; const char* name; };
    const Shape shapes[] = {
      {98500, 2304, 768, 0, 1, 0, "qkv"}, {98500, 3072, 768, 2, 1, 0, "fc1"},
      {98500, 768, 768, 0, 0, 1, "outproj"}, {98500, 768, 3072, 0, 0, 1, "fc2"},
      {4096, 4096, 4096, 0, 1, 0, "4k"}, {8192, 8192, 8192, 0, 1, 0, "8k"},
    };
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
      _Float16 *A, *B; float *bias, *R = nullptr; void* C;
      const long long MN = (long long)sh.M * sh.N;
      HIP_OK(hipMalloc(&A, (size_t)sh.M * sh.K * 2)); HIP_OK(hipMalloc(&B, (size_t)sh.N * sh.K * 2));
      HIP_OK(hipMalloc(&C, MN * (sh.out16 ? 2 : 4))); HIP_OK(hipMalloc(&bias, sh.N * 4));
      fill_f16<<<2048, 256, 0, s>>>(A, (long long)sh.M * sh.K, 11u, 0, time_scale);
      fill_f16<<<2048, 256, 0, s>>>(B, (long long)sh.N * sh.K, 23u, 0, time_scale / sqrtf((float)sh.K));
      fill_f32<<<64, 256, 0, s>>>(bias, sh.N, 5u, 0, time_scale);
      if (sh.res) { HIP_OK(hipMalloc(&R, MN * 4)); fill_f32<<<2048, 256, 0, s>>>(R, MN, 7u, 0, 1.f); }
      std::vector<std::vector<double>> ms(variants.size());
      for (int r = 0; r < rounds + 1; ++r)
        for (size_t vi = 0; vi < variants.size(); ++vi) {
          const Variant& v = variants[vi];
          OVIS_OKAY(ovis_set_f16_gemm_mode(v.mode, v.grp, v.desync));
          ovis_pp_debug(v.dbg & 255, nullptr); ovis_pp_epilogue(v.dbg >> 8);
          HIP_OK(hipEventRecord(e0, s));
          for (int it = 0; it < iters; ++it)
            OVIS_OKAY(ovis_gemm_nt_f16(A, sh.K, B, sh.K, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, sh.act, sh.out16, s));
          HIP_OK(hipEventRecord(e1, s)); HIP_OK(hipEventSynchronize(e1));
          float t; HIP_OK(hipEventElapsedTime(&t, e0, e1));
          if (r > 0) ms[vi].push_back(t / iters);                 // round 0 = warm-up
        }
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        std::sort(ms[vi].begin(), ms[vi].end());
        const double med = ms[vi][ms[vi].size() / 2], mn = ms[vi][0];
        const double fl = 2.0 * sh.M * sh.N * sh.K;
        printf("time %-8s M=%d N=%d K=%d variant=%-10s median %.4f ms (%.0f TF)  min %.4f ms (%.0f TF)\n", sh.name, sh.M, sh.N, sh.K,
               variants[vi].name.c_str(), med, fl / med / 1e9, mn, fl / mn / 1e9);
      }

      if (do_trace)
        for (const Variant& v : variants) {
          if (v.mode != 1) continue;
          const size_t nst = 256 * 16 * 2 * 4 + 256 * 2 * 64;
          unsigned long long* d_st; HIP_OK(hipMalloc(&d_st, nst * 8)); HIP_OK(hipMemsetAsync(d_st, 0, nst * 8, s));
          OVIS_OKAY(ovis_set_f16_gemm_mode(v.mode, v.grp, v.desync));
          ovis_pp_debug(v.dbg & 255, d_st); ovis_pp_epilogue(v.dbg >> 8);
          OVIS_OKAY(ovis_gemm_nt_f16(A, sh.K, B, sh.K, C, sh.N, sh.M, sh.N, sh.K, bias, R, sh.N, sh.act, sh.out16, s));
          ovis_pp_debug(0, nullptr);
          std::vector<unsigned long long> st(nst);
          HIP_OK(hipMemcpyAsync(st.data(), d_st, nst * 8, hipMemcpyDeviceToHost, s)); HIP_OK(hipStreamSynchronize(s));
          unsigned long long t0 = ~0ull, t1 = 0;
          for (size_t i = 0; i < 256 * 16 * 2 * 4; i += 4) if (st[i]) { t0 = std::min(t0, st[i]); t1 = std::max(t1, st[i + 2]); }
          printf("trace %s variant=%s: first tile begin -> last epilogue end %.2f us (100 MHz ticks)\n", sh.name, v.name.c_str(), (t1 - t0) * 0.01);
          {   // in-kernel clock: shader-clock ticks / 100 MHz ticks between the first and the last stamped tile begin of a workgroup (median)
            std::vector<double> ghz;
            for (int b = 0; b < 256; ++b) {
              const unsigned long long* x0 = &st[((size_t)(b * 16 + 0) * 2 + 0) * 4];
              int last = 0;
              for (int it = 1; it < 16; ++it) if (st[((size_t)(b * 16 + it) * 2 + 0) * 4]) last = it;
              const unsigned long long* x1 = &st[((size_t)(b * 16 + last) * 2 + 0) * 4];
              if (last > 0 && x1[0] > x0[0]) ghz.push_back((double)(x1[3] - x0[3]) / (double)(x1[0] - x0[0]) * 0.1);
            }
            if (!ghz.empty()) { std::sort(ghz.begin(), ghz.end()); printf("trace %s variant=%s: in-kernel clock %.3f GHz (median of %zu workgroups; min %.3f max %.3f)\n", sh.name, v.name.c_str(), ghz[ghz.size() / 2], ghz.size(), ghz[0], ghz.back()); }
          }
          for (int it = 0; it < 16; ++it) {
            double kl[2] = {0, 0}, ep[2] = {0, 0}, gap[2] = {0, 0}, beg[2] = {0, 0}; int n[2] = {0, 0}; double bmin = 1e30, bmax = 0;
            for (int b = 0; b < 256; ++b) for (int g = 0; g < 2; ++g) {
              const unsigned long long* x = &st[((size_t)(b * 16 + it) * 2 + g) * 4];
              if (!x[0] || !x[2]) continue;
              kl[g] += (x[1] - x[0]) * 0.01; ep[g] += (x[2] - x[1]) * 0.01; beg[g] += (x[0] - t0) * 0.01; ++n[g];
              bmin = std::min(bmin, (x[0] - t0) * 0.01); bmax = std::max(bmax, (x[0] - t0) * 0.01);
              if (it + 1 < 16) { const unsigned long long* y = &st[((size_t)(b * 16 + it + 1) * 2 + g) * 4]; if (y[0]) gap[g] += (y[0] - x[2]) * 0.01; }
            }
            if (n[0]) printf("  it %2d (%3d wg): begin %.1f us [%.1f..%.1f]  G0 kloop %.2f epi %.2f | G1 kloop %.2f epi %.2f\n", it, n[0], beg[0] / n[0], bmin, bmax,
                             kl[0] / n[0], ep[0] / n[0], kl[1] / std::max(n[1], 1), ep[1] / std::max(n[1], 1));
          }
          {   // per-K-step durations of tile iterations 2..5 (mean over workgroups, group 0): dt[kt] = end(kt) - end(kt-1); dt[0] = end(0) - tile begin
            const int nks = std::min(sh.K / 64, 16);
            for (int it = 2; it < 6; ++it) {
              printf("  ksteps it %d:", it);
              for (int kt = 0; kt < nks; ++kt) {
                double sum = 0; int n = 0;
                for (int b = 0; b < 256; ++b) {
                  const unsigned long long* k = &st[256 * 16 * 2 * 4 + (size_t)(b * 2 + 0) * 64 + (it - 2) * 16];
                  const unsigned long long* x = &st[((size_t)(b * 16 + it) * 2 + 0) * 4];
                  if (!k[kt] || !x[0]) continue;
                  sum += (k[kt] - (kt ? k[kt - 1] : x[0])) * 0.01; ++n;
                }
                printf(" %.2f", n ? sum / n : 0.0);
              }
              printf("\n");
            }
          }
          HIP_OK(hipFree(d_st));
        }
      HIP_OK(hipFree(A)); HIP_OK(hipFree(B)); HIP_OK(hipFree(C)); HIP_OK(hipFree(bias));
      if (R) HIP_OK(hipFree(R));
    }
  }
  printf("gemm_lab done, %d failing checks\n", fails);
  return fails ? 1 : 0;
}
