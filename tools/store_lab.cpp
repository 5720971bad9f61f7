// store_lab: how fast can one CU issue the epilogue stores of a 256x256 GEMM tile, by store shape?
//   hipcc --offload-arch=gfx950 -O2 tools/store_lab.cpp -o build/store_lab
// Every workgroup (512 threads, one per CU) "stores a tile" `iters` times; a wave owns 128 rows x ROWB bytes per tile
// (ROWB = 128: fp16 outputs of 64 columns, 256: f32).  Patterns: rows covered by one 1-KB wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// SEG = contiguous bytes per row per instruction (64, 128, 256); lanes per row = SEG / 16; rows per instr = 1024 / SEG
template <int SEG, int ROWB>
__global__ void __launch_bounds__(512) store_kernel(char* __restrict__ C, long long ldc_bytes, int tiles_n, int n_tiles, int iters, int compute_cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  constexpr int LPR = SEG / 16, RPI = 1024 / SEG;                 // lanes per row, rows per instruction
  const int r = lane / LPR, c = lane % LPR;
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int it = 0; it < iters; ++it) {
    const int tile = (blockIdx.x + it * gridDim.x) % n_tiles;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    char* base = C + ((long long)tm * 256 + wr * 128) * ldc_bytes + ((long long)tn * 4 + wc) * ROWB;
    if (compute_cycles) { const long long t0 = clock64(); while (clock64() - t0 < compute_cycles) {} }
#pragma unroll
    for (int rb = 0; rb < 128 / RPI; ++rb)
#pragma unroll
      for (int s = 0; s < ROWB / SEG; ++s)
        *reinterpret_cast<uint4*>(base + (long long)(rb * RPI + r) * ldc_bytes + s * SEG + c * 16) = v;
  }
}

template <int SEG, int ROWB>
static void run(const char* name, char* C, long long ldc_bytes, int tiles_m, int tiles_n, int compute_cycles, int grid = 256) {
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const int n_tiles = tiles_m * tiles_n, iters = 14;
  for (int rep = 0; rep < 3; ++rep) {
    HIP_OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((store_kernel<SEG, ROWB>), dim3(grid), dim3(512), 0, 0, C, ldc_bytes, tiles_n, n_tiles, iters, compute_cycles);
    HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * iters * 256 * 4 * ROWB;
    if (rep == 2)
      printf("%-22s grid %3d seg %3d B rowbytes %3d compute %6d cyc: %.3f ms  %.2f TB/s  %.1f GB/s/CU  %.2f us per tile\n", name, grid, SEG, ROWB, compute_cycles, ms,
             bytes / ms / 1e9, bytes / ms / 1e6 / grid, ms * 1e3 / iters);
  }
}

int main() {
  const int tiles_m = 385, tiles_n = 9;
  const long long ld16 = 2304 * 2, ld32 = 2304 * 4;
  char* C; HIP_OK(hipMalloc(&C, (size_t)tiles_m * 256 * ld32));
  for (int cc : {0, 30000}) {
    run<64, 128>("fp16 16 rows x 64 B", C, ld16, tiles_m, tiles_n, cc);
    run<128, 128>("fp16 8 rows x 128 B", C, ld16, tiles_m, tiles_n, cc);
    run<64, 256>("f32 16 rows x 64 B", C, ld32, tiles_m, tiles_n, cc);
    run<128, 256>("f32 8 rows x 128 B", C, ld32, tiles_m, tiles_n, cc);
    run<256, 256>("f32 4 rows x 256 B", C, ld32, tiles_m, tiles_n, cc);
  }
  for (int grid : {8, 32, 64, 128}) { run<64, 128>("fp16 16 rows x 64 B", C, ld16, tiles_m, tiles_n, 0, grid); run<128, 128>("fp16 8 rows x 128 B", C, ld16, tiles_m, tiles_n, 0, grid); run<256, 256>("f32 4 rows x 256 B", C, ld32, tiles_m, tiles_n, 0, grid); }
  return 0;
}
