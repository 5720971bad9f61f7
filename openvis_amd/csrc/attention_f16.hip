// fp16-operand flash attention for the CLIP ViT tower (197 x 197 tokens, heads x 64) on gfx950.
//
// Replaces the nn.MultiheadAttention core of mask_adapted_clip/model.py:254-263 on the fp16 CLIP path (the reference's
// GPU CLIP runs in fp16).  Same structure as attention_f32.hip — transposed scores S^T = K Q^T so one query lives on a
// lane and P is directly the B operand of O^T += V^T P^T — with v_mfma_f32_32x32x16_f16 (f32 accumulate, f32 softmax):
//   * K tile [32 keys][64] fp16 in LDS, rows padded to 144 B, fragments by ds_read_b128 (conflict-free);
//   * V tile [32 keys][64] fp16 in LDS, rows padded to 192 B, consumed TRANSPOSED with ds_read_b64_tr_b16
//     (gfx950 hardware transpose read): a 16-lane group fetches a 4-key x 16-d block and each lane receives the
//     4 keys of its own d column — exactly the k-order in which the accumulator registers hold P
//     (register r of lane-half h  <->  key (r&3) + 8(r>>2) + 4h), so P never leaves registers.
#include "common.h"
#include <type_traits>

#ifndef ATT_PAIR
#define ATT_PAIR 0
#endif
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

struct AttnH {
  const _Float16* q; long long q_bs; int q_ld;
  const _Float16* k; long long k_bs; int k_ld;
  const _Float16* v; long long v_bs; int v_ld;
  _Float16* out; long long o_bs; int o_ld;
  int B, H, Nq, Nk;
  float scale;
  int dbg;            // lab only (ovis_attention_f16_debug): 1 = stop after the K/V staging, 2 = skip the K/V loads (compute on whatever LDS holds)
};

// value of the lane 32 away, by v_permlane32_swap (VALU: no LDS round trip in the softmax's dependent chain, unlike ds_bpermute)
__device__ __forceinline__ float max_xor32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_xor32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ f16x4 tr_read(const _Float16* lds_ptr) {
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<_Float16*>(lds_ptr)));
  return __builtin_bit_cast(f16x4, r);
}

__global__ void __launch_bounds__(256)
flash_attn_f16_kernel(AttnH a) {
  constexpr int D = 64;
  constexpr int KROW = D + 8;    // halfs -> 144 B rows (ds_read_b128, conflict-free)
  constexpr int VROW = D + 32;   // halfs -> 192 B rows (4 rows x 64 B of a tr-read group tile the 64 banks)
  __shared__ __attribute__((aligned(16))) _Float16 Ks[32 * KROW];
  __shared__ __attribute__((aligned(16))) _Float16 Vs[32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, head = bh % a.H;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const int qi = q0 + r32;
  const bool wave_active = q0 < a.Nq;
  const bool q_ok = qi < a.Nq;

  const _Float16* qp = a.q + b * a.q_bs + (long long)head * D;
  const _Float16* kp = a.k + b * a.k_bs + (long long)head * D;
  const _Float16* vp = a.v + b * a.v_bs + (long long)head * D;

  // Q fragments (B operand of S^T = K Q^T): lane (q = r32, half h), step s holds d = 16s + 8h .. +8
  f16x8 qf[D / 16];
#pragma unroll
  for (int s = 0; s < D / 16; ++s) {
    const uint4 t = *reinterpret_cast<const uint4*>(qp + (long long)(q_ok ? qi : 0) * a.q_ld + 16 * s + 8 * h);
    const uint4 z = make_uint4(q_ok ? t.x : 0u, q_ok ? t.y : 0u, q_ok ? t.z : 0u, q_ok ? t.w : 0u);
    qf[s] = __builtin_bit_cast(f16x8, z);
  }

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: thread -> (key row = tid/8, 16-byte chunk = tid%8)
  const int srow = tid >> 3, sch = tid & 7;
  uint4 pk, pv;
  bool okr;
  auto gload = [&](int kt) {
    const int key = kt + srow;
    okr = key < a.Nk;
    const int kc = okr ? key : 0;
    pk = *reinterpret_cast<const uint4*>(kp + (long long)kc * a.k_ld + sch * 8);
    pv = *reinterpret_cast<const uint4*>(vp + (long long)kc * a.v_ld + sch * 8);
  };
  auto lstore = [&]() {
    *reinterpret_cast<uint4*>(&Ks[srow * KROW + sch * 8]) = make_uint4(okr ? pk.x : 0u, okr ? pk.y : 0u, okr ? pk.z : 0u, okr ? pk.w : 0u);
    *reinterpret_cast<uint4*>(&Vs[srow * VROW + sch * 8]) = make_uint4(okr ? pv.x : 0u, okr ? pv.y : 0u, okr ? pv.z : 0u, okr ? pv.w : 0u);
  };

  // transposed-read addressing: group g = lane>>4 fetches block rows key0+q (q = (lane&15)>>2), cols dcol0 + 4p (p = lane&3)
  const int g = lane >> 4, li = lane & 15;
  const int tr_off = (li >> 2) * VROW + 16 * (g & 1) + 4 * (li & 3);   // + key0*VROW + t*32

  const float scale_log2e = a.scale * 1.4426950408889634f;
  gload(0);
  for (int kt = 0; kt < a.Nk; kt += 32) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 32 < a.Nk) gload(kt + 32);
    if (!wave_active) continue;

    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < D / 16; ++st) {
      const f16x8 kk = *reinterpret_cast<const f16x8*>(&Ks[r32 * KROW + 16 * st + 8 * h]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk, qf[st], s, 0, 0, 0);
    }
    // unscaled scores: running max on the raw dot products, scale folded into the exponent's fma; only the last key
    // tile can hold keys >= Nk
    if (kt + 32 > a.Nk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        s[r] = key < a.Nk ? s[r] : -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);       // finite: every tile holds at least one valid key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
    const float moff = -m_new * scale_log2e;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2e, moff));
      s[r] = p;
      ps += p;
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

    // O^T += V^T P^T : step sp covers keys 16sp..16sp+15; lane-half h element j <-> key 16sp + 4h + (j&3) + 8(j>>2)
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      f16x8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (_Float16)s[8 * sp + j];
      const int key0a = 16 * sp + 4 * h, key0b = key0a + 8;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x4 va = tr_read(&Vs[key0a * VROW + t * 32 + tr_off]);
        const f16x4 vb = tr_read(&Vs[key0b * VROW + t * 32 + tr_off]);
        f16x8 av;
        av[0] = va[0]; av[1] = va[1]; av[2] = va[2]; av[3] = va[3];
        av[4] = vb[0]; av[5] = vb[1]; av[6] = vb[2]; av[7] = vb[3];
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb, o[t], 0, 0, 0);
      }
    }
  }

  if (!wave_active || !q_ok) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  _Float16* op = a.out + b * a.o_bs + (long long)qi * a.o_ld + head * D;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = t * 32 + 8 * g4 + 4 * h;
      f16x4 r;
      r[0] = (_Float16)(o[t][4 * g4] * inv); r[1] = (_Float16)(o[t][4 * g4 + 1] * inv);
      r[2] = (_Float16)(o[t][4 * g4 + 2] * inv); r[3] = (_Float16)(o[t][4 * g4 + 3] * inv);
      *reinterpret_cast<f16x4*>(op + d) = r;
    }
}

// Whole-sequence variant for short sequences (Nk <= 224, e.g. the 197 tokens of ViT-B/16 @224): one workgroup = one
// (batch, head), one wavefront = one tile of 32 queries; K and V of the head are staged ONCE in LDS (the tiled kernel
// above re-stages them for every group of 4 query tiles and synchronises twice per key tile), then every wavefront
// walks the key tiles on its own with the online-softmax recurrence -- no barrier inside the loop.
// G = 8-key groups of the LAST key tile that hold a valid key (Nk - 32 (KT - 1) <= 8 G).  The score registers of a lane cover the tile's
// keys in groups of 8 (register r: key (r & 3) + 8 (r >> 2) + 4 h), so with G < 4 the max / exp / sum work of the padded groups and the
// P V products of the padded 16-key halves are skipped outright instead of being computed on -inf scores (197 tokens = 6 tiles + 5 keys:
// G = 1 drops 3/4 of the last tile's softmax work and half of its P V MFMAs; same values, the skipped terms were exact zeros).
template <int KT, int G = 4>
__global__ void __launch_bounds__(64 * KT, 4)          // <= 128 VGPRs: two 7-wavefront workgroups per CU (their LDS: 2 x 75 KB)
flash_attn_f16_seq_kernel(AttnH a) {
  constexpr int D = 64;
  constexpr int KROW = D + 8;    // halfs -> 144 B rows (ds_read_b128, conflict-free)
  constexpr int VROW = D + 32;   // halfs -> 192 B rows (transposed reads tile the 64 banks)
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  _Float16* Ks = smem;
  _Float16* Vs = smem + 32 * KT * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int head = blockIdx.x, b = blockIdx.y;
  const _Float16* qp = a.q + b * a.q_bs + (long long)head * D;
  const _Float16* kp = a.k + b * a.k_bs + (long long)head * D;
  const _Float16* vp = a.v + b * a.v_bs + (long long)head * D;

  const int qi = wave * 32 + r32;
  const bool q_ok = qi < a.Nq;
  const int qc = q_ok ? qi : a.Nq - 1;
  f16x8 qf[D / 16];
#pragma unroll
  for (int st = 0; st < D / 16; ++st) qf[st] = *reinterpret_cast<const f16x8*>(qp + (long long)qc * a.q_ld + 16 * st + 8 * h);
  // K / V staging: 32*KT rows x 8 chunks of 16 B over 64*KT threads = exactly 4 chunks each.  All 8 loads (and the 4 Q
  // loads above) are issued before the first LDS store -- as a rolled loop this was 4 dependent memory round trips
  // (load, wait, store) per workgroup.
  uint4 kk[4], vv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + it * 64 * KT;
    const int row = c >> 3, ch = c & 7;
    const int rc = row < a.Nk ? row : 0;
    if (a.dbg == 2) { kk[it] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); vv[it] = kk[it]; continue; }
    kk[it] = *reinterpret_cast<const uint4*>(kp + (long long)rc * a.k_ld + ch * 8);
    vv[it] = *reinterpret_cast<const uint4*>(vp + (long long)rc * a.v_ld + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {                                // rows >= Nk zero
    const int c = tid + it * 64 * KT;
    const int row = c >> 3, ch = c & 7;
    const bool ok = row < a.Nk;
    *reinterpret_cast<uint4*>(&Ks[row * KROW + ch * 8]) =
        make_uint4(ok ? kk[it].x : 0u, ok ? kk[it].y : 0u, ok ? kk[it].z : 0u, ok ? kk[it].w : 0u);
    *reinterpret_cast<uint4*>(&Vs[row * VROW + ch * 8]) =
        make_uint4(ok ? vv[it].x : 0u, ok ? vv[it].y : 0u, ok ? vv[it].z : 0u, ok ? vv[it].w : 0u);
  }
  __syncthreads();
  if (wave * 32 >= a.Nq) return;
  if (a.dbg == 1) { if (q_ok && lane == 0) a.out[b * a.o_bs + (long long)qi * a.o_ld + head * D] = Ks[lane]; return; }

  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (li >> 2) * VROW + 16 * (g16 & 1) + 4 * (li & 3);
  const float sl2 = a.scale * 1.4426950408889634f;
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  constexpr int nkt = KT;                                        // the dispatch picks KT = ceil(Nk / 32)
  // Key tiles are walked in PAIRS (64 keys per online-softmax step; a single tile closes an odd count): one running-max update, one
  // alpha and one rescale of the 32 output registers per 64 keys instead of per 32 -- the loop is VALU-bound (softmax), not MFMA-bound.
  auto step = [&](auto nt_tag, int kt0) {
    constexpr int NT = decltype(nt_tag)::value;
    f32x16 s[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      // two independent accumulation chains of two MFMAs each (a chain of four waits out the 16-pass latency three times)
      f32x16 s2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[u][r] = 0.f; s2[r] = 0.f; }
      f16x8 kk[D / 16];
#pragma unroll
      for (int st = 0; st < D / 16; ++st) kk[st] = *reinterpret_cast<const f16x8*>(&Ks[((kt0 + u) * 32 + r32) * KROW + 16 * st + 8 * h]);
      s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk[0], qf[0], s[u], 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk[2], qf[2], s2, 0, 0, 0);
      s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk[1], qf[1], s[u], 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk[3], qf[3], s2, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] += s2[r];
    }
    // scores stay unscaled: the running max is tracked on the raw dot products (scale > 0 keeps the order) and the scale
    // is folded into the exponent's fma; only the last key tile can hold keys >= Nk, so only it is masked
    const bool last = kt0 + NT == nkt;                             // (nkt == KT by the dispatch: decided when the loop below is unrolled)
    auto nreg = [&](int u) { return (last && u == NT - 1) ? 4 * G : 16; };   // score registers of tile u that can hold a valid key
    if (last) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r >= 4 * G) continue;
        const int key = (kt0 + NT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        s[NT - 1][r] = key < a.Nk ? s[NT - 1][r] : -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r < nreg(u)) mt = fmaxf(mt, s[u][r]);
    mt = max_xor32(mt);
    const float m_new = fmaxf(m_run, mt);                        // finite: every key tile holds at least one valid key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
    const float moff = -m_new * sl2;
    float ps = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < nreg(u)) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], sl2, moff));
          s[u][r] = p;
          ps += p;
        } else {
          s[u][r] = 0.f;                                           // a padded group: exp2(-inf) = 0 exactly
        }
      }
    ps = sum_xor32(ps);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        if (8 * sp >= nreg(u)) continue;                           // a 16-key half of padded keys only: P = 0, nothing to add
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
        f16x8 pb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                              // packed conversion (v_cvt_pk_f16_f32), round to nearest even
          const f16x2 hp = __builtin_convertvector(f32x2{s[u][8 * sp + 2 * j], s[u][8 * sp + 2 * j + 1]}, f16x2);
          pb[2 * j] = hp[0]; pb[2 * j + 1] = hp[1];
        }
        const int key0a = (kt0 + u) * 32 + 16 * sp + 4 * h, key0b = key0a + 8;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f16x4 va = tr_read(&Vs[key0a * VROW + t * 32 + tr_off]);
          const f16x4 vb = tr_read(&Vs[key0b * VROW + t * 32 + tr_off]);
          f16x8 av;
          av[0] = va[0]; av[1] = va[1]; av[2] = va[2]; av[3] = va[3];
          av[4] = vb[0]; av[5] = vb[1]; av[6] = vb[2]; av[7] = vb[3];
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb, o[t], 0, 0, 0);
        }
      }
  };
#pragma unroll
  for (int kt = 0; kt < KT; kt += 2) {
    if (kt >= nkt) break;
    if (ATT_PAIR && kt + 1 < KT && kt + 1 < nkt) step(std::integral_constant<int, 2>{}, kt);
    else { step(std::integral_constant<int, 1>{}, kt); if (!ATT_PAIR && kt + 1 < KT && kt + 1 < nkt) step(std::integral_constant<int, 1>{}, kt + 1); }
  }
  if (!q_ok) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  _Float16* op = a.out + b * a.o_bs + (long long)qi * a.o_ld + head * D;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = t * 32 + 8 * g4 + 4 * h;
      f16x4 r;
      r[0] = (_Float16)(o[t][4 * g4] * inv); r[1] = (_Float16)(o[t][4 * g4 + 1] * inv);
      r[2] = (_Float16)(o[t][4 * g4 + 2] * inv); r[3] = (_Float16)(o[t][4 * g4 + 3] * inv);
      *reinterpret_cast<f16x4*>(op + d) = r;
    }
}

template <int KT, int G = 4>
int launch_seq(const AttnH& a, hipStream_t stream) {
  const size_t shmem = (size_t)32 * KT * ((64 + 8) + (64 + 32)) * sizeof(_Float16);
  static bool attr_set = false;
  if (shmem > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_f16_seq_kernel<KT, G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess)
      return ovis::fail(OVIS_EINVAL, "attention_f16: cannot raise the dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL((flash_attn_f16_seq_kernel<KT, G>), dim3(a.H, a.B), dim3(64 * KT), shmem, stream, a);
  return ovis::check_launch("attention_f16 (whole sequence)");
}

}  // namespace

static int g_attn_dbg = 0;
extern "C" int ovis_attention_f16_debug(int m) { g_attn_dbg = m; return OVIS_OK; }   // lab only

extern "C" int ovis_attention_f16(const void* q, long long q_bs, int q_ld, const void* k, long long k_bs, int k_ld,
                                  const void* v, long long v_bs, int v_ld, void* out, long long o_bs, int o_ld, int B,
                                  int H, int Nq, int Nk, int D, float scale, ovis_stream_t stream) {
  OVIS_REQUIRE(q && k && v && out, "attention_f16: null pointer");
  OVIS_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attention_f16: non-positive size");
  OVIS_REQUIRE(D == 64, "attention_f16: head dim %d not supported (64)", D);
  OVIS_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && v_ld % 8 == 0 && o_ld % 4 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 &&
                   v_bs % 8 == 0 && o_bs % 4 == 0,
               "attention_f16: strides must be multiples of 8 halfs");
  OVIS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)out & 7) == 0,
               "attention_f16: q/k/v must be 16-byte aligned");
  AttnH a;
  a.q = (const _Float16*)q; a.q_bs = q_bs; a.q_ld = q_ld; a.k = (const _Float16*)k; a.k_bs = k_bs; a.k_ld = k_ld;
  a.v = (const _Float16*)v; a.v_bs = v_bs; a.v_ld = v_ld; a.out = (_Float16*)out; a.o_bs = o_bs; a.o_ld = o_ld;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale; a.dbg = g_attn_dbg;
  // short sequences: whole K/V of a head in LDS, one wavefront per query tile
  const int kt = ovis::cdiv(Nk, 32);
  if (kt <= 7 && ovis::cdiv(Nq, 32) <= kt) {
    switch (kt) {
      case 1: return launch_seq<1>(a, (hipStream_t)stream);
      case 2: return launch_seq<2>(a, (hipStream_t)stream);
      case 3: return launch_seq<3>(a, (hipStream_t)stream);
      case 4: return launch_seq<4>(a, (hipStream_t)stream);
      case 5: return launch_seq<5>(a, (hipStream_t)stream);
      case 6: return launch_seq<6>(a, (hipStream_t)stream);
      default:                                        // 7 tiles: the 197 tokens of ViT-B/16 @224 leave 5 keys in the last one
        switch (ovis::cdiv(Nk - 192, 8)) {
          case 1: return launch_seq<7, 1>(a, (hipStream_t)stream);
          case 2: return launch_seq<7, 2>(a, (hipStream_t)stream);
          case 3: return launch_seq<7, 3>(a, (hipStream_t)stream);
          default: return launch_seq<7, 4>(a, (hipStream_t)stream);
        }
    }
  }
  hipLaunchKernelGGL(flash_attn_f16_kernel, dim3(ovis::cdiv(Nq, 128), B * H), dim3(256), 0, (hipStream_t)stream, a);
  return ovis::check_launch("attention_f16");
}
