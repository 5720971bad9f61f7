// COCO run-length encoding of binary masks on the GPU (the evaluator hand-off of
// openvis/data/evals/ytvis_eval.py:258-301, which copies every [H,W] mask to the host and calls pycocotools' rleEncode).
//
// Input: n masks of `len` bytes each, already in the order RLE scans them (column-major: ovis_final_masks_u8 with
// column_major = 1).  Output per mask: the uncompressed COCO counts (run lengths alternating 0-run, 1-run, ...; the
// first count is the number of leading zeros and may be 0) and their number.  One workgroup per mask walks it in
// chunks of 1024 x 16 bytes: every thread counts the value changes inside its 16 bytes (the byte before them comes
// from its left neighbour / the previous chunk), a workgroup scan gives each change its index, change POSITIONS are
// written, and a second pass turns positions into run lengths.
#include "common.h"

namespace {

constexpr int RLE_THREADS = 1024;
constexpr int RLE_PER_THREAD = 16;

__global__ void __launch_bounds__(RLE_THREADS)
rle_encode_kernel(const uint8_t* __restrict__ masks, long long len, int* __restrict__ counts, int* __restrict__ n_runs, int cap) {
  __shared__ int wave_sum[RLE_THREADS / 64];
  __shared__ uint8_t last_byte[RLE_THREADS];
  __shared__ int s_offset, s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint8_t* m = masks + (long long)blockIdx.x * len;
  int* pos = counts + (long long)blockIdx.x * cap;
  if (tid == 0) { s_offset = 0; s_carry = 0; }
  __syncthreads();
  const long long chunk = (long long)RLE_THREADS * RLE_PER_THREAD;
  for (long long base = 0; base < len; base += chunk) {
    const long long i0 = base + (long long)tid * RLE_PER_THREAD;
    uint8_t v[RLE_PER_THREAD];
    const bool vec = ((reinterpret_cast<uintptr_t>(m + i0) & 15) == 0) && i0 + RLE_PER_THREAD <= len;
    if (vec) {
      const uint4 q = *reinterpret_cast<const uint4*>(m + i0);
      const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < RLE_PER_THREAD; ++e) v[e] = (wds[e >> 2] >> (8 * (e & 3))) & 0xffu ? 1 : 0;
    } else {
#pragma unroll
      for (int e = 0; e < RLE_PER_THREAD; ++e) v[e] = (i0 + e < len) ? (m[i0 + e] ? 1 : 0) : 2;     // 2 = beyond the end
    }
    // last valid byte of this thread (threads past the end repeat their predecessor's view via the carry chain)
    int n_valid = (int)(len - i0 < RLE_PER_THREAD ? (len - i0 > 0 ? len - i0 : 0) : RLE_PER_THREAD);
    last_byte[tid] = n_valid > 0 ? v[n_valid - 1] : 255;
    __syncthreads();
    int prev;                                           // value before this thread's first byte
    if (tid == 0) prev = s_carry;
    else {
      int t = tid - 1;                                  // nearest thread to the left that holds valid bytes (always t: chunks are dense)
      prev = last_byte[t];
    }
    int cnt = 0;
    int p = prev;
#pragma unroll
    for (int e = 0; e < RLE_PER_THREAD; ++e)
      if (e < n_valid) { cnt += (v[e] != p); p = v[e]; }
    // workgroup exclusive scan of cnt
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o, 64);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int wv = 0; wv < RLE_THREADS / 64; ++wv) {
      const int sv = wave_sum[wv];
      if (wv < wave) wave_off += sv;
      total += sv;
    }
    int o = s_offset + wave_off + incl - cnt;
    p = prev;
#pragma unroll
    for (int e = 0; e < RLE_PER_THREAD; ++e)
      if (e < n_valid) {
        if (v[e] != p) { if (o < cap) pos[o] = (int)(i0 + e); ++o; }
        p = v[e];
      }
    __syncthreads();
    if (tid == 0) {
      s_offset += total;
      const long long end = base + chunk < len ? base + chunk : len;
      s_carry = m[end - 1] ? 1 : 0;
    }
    __syncthreads();
  }
  // positions -> run lengths: counts[0] = pos[0], counts[j] = pos[j] - pos[j-1], counts[k] = len - pos[k-1]
  const int k = s_offset;                               // number of value changes (the implicit value before the mask is 0)
  if (tid == 0) n_runs[blockIdx.x] = k + 1;
  const int kk = k < cap ? k : cap - 1;                 // overflow: n_runs > cap tells the caller the buffer was too small
  // slabs from the top down: counts[j] needs pos[j-1], which lives in the slab below and must not be overwritten yet
  for (int j0 = kk / RLE_THREADS * RLE_THREADS; j0 >= 0; j0 -= RLE_THREADS) {
    const int j = j0 + tid;
    int val = 0;
    const bool ok = j <= kk;
    if (ok) {
      const int hi = j < k ? pos[j] : (int)len;
      const int lo = j > 0 ? pos[j - 1] : 0;
      val = hi - lo;
    }
    __syncthreads();                                    // all reads of this slab (and pos[j0-1]) done before overwriting
    if (ok) pos[j] = val;
    __syncthreads();
  }
}

}  // namespace

extern "C" int ovis_rle_encode_u8(const uint8_t* masks, int n_masks, long long len, int* counts, int* n_runs, int cap,
                                  ovis_stream_t stream) {
  OVIS_REQUIRE(masks && counts && n_runs, "rle_encode: null pointer");
  OVIS_REQUIRE(n_masks > 0 && len > 0 && len < (1ll << 31) && cap >= 2, "rle_encode: bad sizes");
  hipLaunchKernelGGL(rle_encode_kernel, dim3(n_masks), dim3(RLE_THREADS), 0, (hipStream_t)stream, masks, len, counts, n_runs, cap);
  return ovis::check_launch("rle_encode");
}
