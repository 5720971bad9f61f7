// Swin window attention (backbone/swin.py:130-169) for gfx950, fp16 operands / f32 softmax -- the reference's autocast
// arithmetic.  One workgroup = one (window, head); one wavefront = one tile of 32 queries; the whole window fits on chip:
//   * K [N,32] and V [N,32] of the head are staged once in LDS (N <= 160 tokens, 20 KB);
//   * S^T = K Q^T (v_mfma_f32_32x32x16_f16): a lane owns ONE query; register r of lane-half h of key tile kt  <->
//     key 32 kt + (r&3) + 8 (r>>2) + 4 h, so the softmax statistics are in-register reductions plus one cross-half
//     shuffle; tiles of 32 keys are folded in with the online (running max / sum) recurrence to keep ~80 VGPRs;
//   * relative-position bias (f32 table [heads,N,ld]) and the shifted-window mask (u8 table [nW,N,ld], window = blockIdx
//     mod nW) are added as 16-byte / 4-byte row loads that match the register layout;
//   * O^T += V^T P^T with V consumed through ds_read_b64_tr_b16 (hardware transpose read), P straight from registers.
// The reference adds -100 for masked pairs; here they get -inf (difference exp(-100) ~ 4e-44 of the softmax mass).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ f16x4 tr_read(const _Float16* lds_ptr) {
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<_Float16*>(lds_ptr)));
  return __builtin_bit_cast(f16x4, r);
}

template <int KT>
__global__ void __launch_bounds__(64 * KT)
swin_window_attn_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, const float* __restrict__ bias,
                        const uint8_t* __restrict__ mask, int N, int C, int nW, int ld, float scale) {
  constexpr int D = 32;
  constexpr int KROW = D + 8;    // halfs: 80-byte rows, conflict-free ds_read_b128
  constexpr int VROW = D;        // halfs: 64-byte rows -- the 4 rows x 64 B of a transposed read tile the 64 banks
  __shared__ __attribute__((aligned(16))) _Float16 Ks[32 * KT * KROW];
  __shared__ __attribute__((aligned(16))) _Float16 Vs[32 * KT * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int head = blockIdx.x;
  const long long win = blockIdx.y;
  const _Float16* base = qkv + win * N * 3 * C + head * D;       // q at +0, k at +C, v at +2C; row stride 3C

  // ---- stage K and V (rows >= N zero) ------------------------------------------------------------
  for (int c = tid; c < 32 * KT * 4; c += 64 * KT) {
    const int row = c >> 2, ch = c & 3;
    const bool ok = row < N;
    const _Float16* p = base + (long long)(ok ? row : 0) * 3 * C + ch * 8;
    const uint4 kk = *reinterpret_cast<const uint4*>(p + C);
    const uint4 vv = *reinterpret_cast<const uint4*>(p + 2 * C);
    *reinterpret_cast<uint4*>(&Ks[row * KROW + ch * 8]) = make_uint4(ok ? kk.x : 0u, ok ? kk.y : 0u, ok ? kk.z : 0u, ok ? kk.w : 0u);
    *reinterpret_cast<uint4*>(&Vs[row * VROW + ch * 8]) = make_uint4(ok ? vv.x : 0u, ok ? vv.y : 0u, ok ? vv.z : 0u, ok ? vv.w : 0u);
  }
  // ---- Q fragments of this wave's query tile ------------------------------------------------------
  const int qi = wave * 32 + r32;
  const bool q_ok = qi < N;
  const int qc = q_ok ? qi : N - 1;
  f16x8 qf[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) qf[st] = *reinterpret_cast<const f16x8*>(base + (long long)qc * 3 * C + 16 * st + 8 * h);
  __syncthreads();

  // ---- per key tile: S^T = K Q^T, + bias / mask, online softmax, O^T += V^T P^T -------------------
  const float* brow = bias + ((long long)head * N + qc) * ld;
  const uint8_t* mrow = mask ? mask + ((win % nW) * N + qc) * (long long)ld : nullptr;
  const float sl2 = scale * 1.4426950408889634f, l2e = 1.4426950408889634f;
  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (li >> 2) * VROW + 16 * (g16 & 1) + 4 * (li & 3);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const f16x8 kk = *reinterpret_cast<const f16x8*>(&Ks[(kt * 32 + r32) * KROW + 16 * st + 8 * h]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk, qf[st], s, 0, 0, 0);
    }
    float mt = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int key0 = kt * 32 + 8 * g + 4 * h;
      const int kc = key0 < ld ? key0 : 0;                       // ld % 4 == 0: a 4-key group is all-in or all-out
      const float4 b = *reinterpret_cast<const float4*>(brow + kc);
      const unsigned mk = mrow ? *reinterpret_cast<const unsigned*>(mrow + kc) : 0u;
      const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool dead = key0 + e >= N || ((mk >> (8 * e)) & 0xffu);
        const float x = dead ? -INFINITY : s[4 * g + e] * sl2 + bb[e] * l2e;
        s[4 * g + e] = x;
        mt = fmaxf(mt, x);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));                        // the other lane-half holds the other keys
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;         // a whole tile can be masked for this row
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - m_safe);
      s[r] = p;
      ps += p;
    }
    ps += __shfl_xor(ps, 32, 64);
    sum = sum * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      f16x8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (_Float16)s[8 * sp + j];
      const int key0a = kt * 32 + 16 * sp + 4 * h, key0b = key0a + 8;
      const f16x4 va = tr_read(&Vs[key0a * VROW + tr_off]);
      const f16x4 vb = tr_read(&Vs[key0b * VROW + tr_off]);
      f16x8 av;
      av[0] = va[0]; av[1] = va[1]; av[2] = va[2]; av[3] = va[3];
      av[4] = vb[0]; av[5] = vb[1]; av[6] = vb[2]; av[7] = vb[3];
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb, o, 0, 0, 0);
    }
  }

  if (!q_ok) return;
  const float inv = 1.f / sum;
  _Float16* op = out + (win * N + qi) * C + head * D;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    f16x4 r;
    r[0] = (_Float16)(o[4 * g4] * inv); r[1] = (_Float16)(o[4 * g4 + 1] * inv);
    r[2] = (_Float16)(o[4 * g4 + 2] * inv); r[3] = (_Float16)(o[4 * g4 + 3] * inv);
    *reinterpret_cast<f16x4*>(op + 8 * g4 + 4 * h) = r;
  }
}

}  // namespace

extern "C" int ovis_swin_window_attention_f16(const void* qkv, void* out, const float* bias, const uint8_t* mask, long long nwin,
                                              int N, int C, int heads, int nW, int ld, float scale, ovis_stream_t stream) {
  OVIS_REQUIRE(qkv && out && bias, "swin_window_attention: null pointer");
  OVIS_REQUIRE(nwin > 0 && nwin < (1ll << 31) && N > 0 && N <= 160 && heads > 0 && C == heads * 32,
               "swin_window_attention: need head_dim 32 and at most 160 tokens per window (N=%d C=%d heads=%d)", N, C, heads);
  OVIS_REQUIRE(ld % 4 == 0 && ld >= N && (!mask || nW > 0), "swin_window_attention: bias/mask rows must be padded to a multiple of 4");
  OVIS_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)bias) & 15) == 0 && (!mask || ((uintptr_t)mask & 3) == 0),
               "swin_window_attention: 16-byte alignment");
  const int KT = (N + 31) / 32;
  const dim3 grid(heads, (unsigned)nwin);
  hipStream_t s = (hipStream_t)stream;
  const _Float16* q = (const _Float16*)qkv;
  _Float16* o = (_Float16*)out;
#define SWIN_ATTN(KT_) hipLaunchKernelGGL((swin_window_attn_kernel<KT_>), grid, dim3(64 * KT_), 0, s, q, o, bias, mask, N, C, nW > 0 ? nW : 1, ld, scale)
  switch (KT) {
    case 1: SWIN_ATTN(1); break;
    case 2: SWIN_ATTN(2); break;
    case 3: SWIN_ATTN(3); break;
    case 4: SWIN_ATTN(4); break;
    default: SWIN_ATTN(5); break;
  }
#undef SWIN_ATTN
  return ovis::check_launch("swin_window_attention");
}
