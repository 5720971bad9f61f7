// Shared GEMM epilogue for the MFMA kernels (gfx950).
//
// All GEMM kernels issue their MFMAs with the operand roles SWAPPED (weights tile as the A operand, activation tile
// as the B operand), so a 32x32 accumulator tile holds ONE output row m per lane (column index of D = lane&31) and
// 4 groups of 4 CONSECUTIVE output columns n in its 16 registers (n = 8g + 4h + e).  The epilogue is then
// 4 x {16-byte bias load, 16-byte residual load, activation, one 16-byte (f32) or 8-byte (fp16) store} per tile
// instead of 16 scattered scalar stores — the store tail was issue-bound, and QuickGELU used a full-precision expf.
#pragma once
#include <hip/hip_runtime.h>

namespace ovis {

using f32x16_t = __attribute__((ext_vector_type(16))) float;
using f16x4_t = __attribute__((ext_vector_type(4))) _Float16;

__device__ __forceinline__ float quick_gelu(float v) {
  // v * sigmoid(1.702 v) with hardware exp2 / rcp (rel. error ~1e-6, far below fp16/f32-accumulate noise of the GEMM)
  return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
}

__device__ __forceinline__ float gelu_erf(float v) {   // nn.GELU() (exact erf form) of the Swin MLP
  return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
}

// Accumulator start value bias + residual of one 32x32 tile (instead of 0) for kernels whose operands are fp16-rounded
// anyway: every residual load of the lane is issued up front and the epilogue (called with bias = R = nullptr) is left
// with conversions and stores.  Only for tiles on epilogue_tile's vector path: 16-byte aligned rows, tile inside N;
// m must be clamped to a valid row by the caller.
__device__ __forceinline__ void acc_init_tile(f32x16_t& acc, long long m, int n_tile0, int h, const float* __restrict__ bias,
                                              const float* __restrict__ R, long long ldr) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n_tile0 + 8 * g + 4 * h;
    const float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 r = R ? *reinterpret_cast<const float4*>(R + m * ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[4 * g + 0] = b.x + r.x; acc[4 * g + 1] = b.y + r.y; acc[4 * g + 2] = b.z + r.z; acc[4 * g + 3] = b.w + r.w;
  }
}

__device__ __forceinline__ bool epilogue_vec_ok(const void* C, long long ldc, const float* bias, const float* R, long long ldr) {
  return ((ldc & 3) == 0) && (!R || (ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
         (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
}

// acc: tile with lane -> row m (m_ok), registers -> columns n_tile0 + 8g + 4h + e.
template <bool OUT_F16>
__device__ __forceinline__ void epilogue_tile(const f32x16_t& acc, long long m, bool m_ok, int n_tile0, int h, int N,
                                              void* __restrict__ C, long long ldc, const float* __restrict__ bias,
                                              const float* __restrict__ R, long long ldr, int act, bool vec_ok) {
  if constexpr (OUT_F16) {
    // fp16 fast path (no residual, whole 32-column tile inside N, 16-byte aligned rows): the two lanes that share an
    // output row (h = 0 / 1) each hold 4 of the 8 columns of a group; v_permlane32_swap exchanges the halves of two
    // groups so that a lane owns 8 consecutive columns and the tile goes out as 16-byte stores (half as many as below)
    if (vec_ok && !R && n_tile0 + 31 < N && (ldc & 7) == 0) {
      unsigned pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if (bias) {
          const float4 b = *reinterpret_cast<const float4*>(bias + n_tile0 + 8 * g + 4 * h);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        } else if (act == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        f16x4_t o;
        o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
        const uint2 u = __builtin_bit_cast(uint2, o);
        pk[g][0] = u.x; pk[g][1] = u.y;
      }
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        // A' = {h=0: A (group gp, cols 0-3), h=1: B from h=0 (group gp+1, cols 0-3)}; B' = {h=0: A from h=1 (gp, cols 4-7), h=1: B}
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[gp][0], pk[gp + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[gp][1], pk[gp + 1][1], false, false);
        if (m_ok) {
          const int n = n_tile0 + 8 * (gp + h);
          *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(C) + m * ldc + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n_tile0 + 8 * g + 4 * h;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (vec_ok && n + 3 < N) {
      if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (R) {
        const float4 r = *reinterpret_cast<const float4*>(R + (m_ok ? m : 0) * ldr + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      if (act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
      } else if (act == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if (m_ok) {
        if constexpr (OUT_F16) {
          f16x4_t o;
          o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
          *reinterpret_cast<f16x4_t*>(reinterpret_cast<_Float16*>(C) + m * ldc + n) = o;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ne = n + e;
        const bool ok = m_ok && ne < N;
        const int nc = ne < N ? ne : 0;
        float x = v[e] + (bias ? bias[nc] : 0.f) + (R ? R[(m_ok ? m : 0) * ldr + nc] : 0.f);
        if (act == 1) x = fmaxf(x, 0.f);
        else if (act == 2) x = quick_gelu(x);
        else if (act == 3) x = gelu_erf(x);
        if (ok) {
          if constexpr (OUT_F16) reinterpret_cast<_Float16*>(C)[m * ldc + ne] = (_Float16)x;
          else reinterpret_cast<float*>(C)[m * ldc + ne] = x;
        }
      }
    }
  }
}

}  // namespace ovis
