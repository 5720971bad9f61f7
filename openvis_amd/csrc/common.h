// Shared helpers for libopenvis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include "../../include/openvis_hip.h"

namespace ovis {

char* err_buf();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OVIS_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return OVIS_OK;
}

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous chunk of the logical grid -> neighbouring tiles hit the same L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// "fp16x2" arithmetic of a large f32 GEMM (gemm_f16_pp.hip FH, gemm_f32x3.h FH): activations are multiplied by a_scale while they are split
// into fp16 hi / lo, the weight planes hold fp16 hi / lo of w * w_scale (both powers of two); flag (device int, may be NULL) is set to 1
// when a result is not finite, i.e. an operand left the fp16 range
struct F16x2 { float a_scale, w_scale; int* flag; };

}  // namespace ovis

#define OVIS_REQUIRE(cond, ...) \
  do { if (!(cond)) return ovis::fail(OVIS_EINVAL, __VA_ARGS__); } while (0)
