// Flash-style multi-head attention on the gfx950 f32 matrix cores.
//
// One kernel serves the three attention shapes of the path:
//   * masked cross-attention of the Mask2Former decoder (video decoder:110-122, 417-426):
//       100 queries x up to T*H_l*W_l keys, 8 heads x 32, boolean mask shared by all heads;
//   * query self-attention (video decoder:52-62): 100 x 100, 8 heads x 32;
//   * CLIP ViT self-attention (mask_adapted_clip/model.py:254-263): 197 x 197, 12 heads x 64, batch = crops.
// Replaces nn.MultiheadAttention's softmax(QK^T/sqrt(d) + mask)V; the in/out projections are GEMMs
// (gemm_f32.hip).  The boolean mask is never replicated per head (the reference repeats it x8,
// video decoder:468) and the score matrix is never materialised.
//
// MI355X mapping (v_mfma_f32_32x32x2_f32, 64-wide wavefronts)
//   * A workgroup = 4 wavefronts = 4 query tiles of 32 rows sharing each staged 32-key K/V tile in LDS
//     (the decoder's 100 queries fit one workgroup); grid = (query groups, batch*heads, KV splits).
//   * Scores are computed TRANSPOSED, S^T = K Q^T: the accumulator then holds one query per lane
//     (column = lane&31) with its 32 keys spread over 16 registers x 2 lane halves, so
//       - the softmax row reductions are 15 in-register max/adds + one cross-half shuffle,
//       - the probabilities are ALREADY the B operand of the next MFMA chain O^T += V^T P^T
//         (register s of lane-half h is key (s&3) + 8(s>>2) + 4h): no LDS round trip, no transposes.
//   * Q fragments stay in registers for the whole KV sweep; K rows are read with ds_read_b128 from
//     rows padded to D+4 floats (conflict-free), V with unit-stride ds_read_b32.
//   * Long KV ranges are split across workgroups (flash-decoding); partial (O, m, l) are merged by
//     attn_combine_kernel.  Rows whose mask blocks every key are treated as unmasked (video decoder:419).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct AttnArgs {
  const float* q; long long q_bs; int q_ld;       // batch stride (elements), row stride
  const float* k; long long k_bs; int k_ld;
  const float* v; long long v_bs; int v_ld;
  void* out; long long o_bs; int o_ld; int out_f16;
  const uint8_t* mask; long long mask_ld;          // [Nq][mask_ld], 1 = blocked; may be null
  long long mask_bs;                               // batch stride of mask (bytes) / of row_open (x Nq); 0 = shared
  const float* bias; long long bias_bs, bias_hs; int bias_ld;   // additive f32 bias [B][H][Nq][bias_ld]; may be null
  const int* row_open;                             // [Nq] number of unblocked keys; may be null
  float* part_o; float* part_ml;                   // split-KV workspace
  int B, H, Nq, Nk, nsplit, keys_per_split;
  float scale;
};

// WAVES = 4: 128 queries per workgroup.  WAVES = 5 (160 queries): for 128 < Nq <= 160 -- Swin's 12x12 windows have 144 tokens, and a
// second 128-query workgroup per (window, head) would stage every K / V tile again for 16 queries.  Threads 0-255 do the staging.
template <int D, int WAVES = 4>
__global__ void __launch_bounds__(WAVES * 64)
flash_attn_f32_kernel(AttnArgs a) {
  constexpr int LDK = D + 4;
  constexpr int DT = D / 32;        // 32-wide output tiles
  constexpr int HD = D / 2;         // d-range owned by a lane half
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, head = bh % a.H;
  const int split = blockIdx.z;
  const int q0 = (blockIdx.x * WAVES + wave) * 32;
  const int qi = q0 + r32;
  const bool wave_active = q0 < a.Nq;
  const bool q_ok = qi < a.Nq;

  const float* qp = a.q + b * a.q_bs + (long long)head * D;
  const float* kp = a.k + b * a.k_bs + (long long)head * D;
  const float* vp = a.v + b * a.v_bs + (long long)head * D;

  // Q fragment: lane (q = r32, half h) keeps d in [h*HD, (h+1)*HD)
  float qf[HD];
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q_ok) t = *reinterpret_cast<const float4*>(qp + (long long)qi * a.q_ld + h * HD + i * 4);
    qf[i * 4 + 0] = t.x; qf[i * 4 + 1] = t.y; qf[i * 4 + 2] = t.z; qf[i * 4 + 3] = t.w;
  }

  const int* ropen = a.row_open ? a.row_open + (a.mask_bs ? (long long)b * a.Nq : 0) : nullptr;
  const bool use_mask = a.mask != nullptr && q_ok && (ropen == nullptr || ropen[qi] > 0);
  const uint8_t* mrow = a.mask ? a.mask + b * a.mask_bs + (long long)(q_ok ? qi : 0) * a.mask_ld : nullptr;

  const float* brow = a.bias ? a.bias + b * a.bias_bs + head * a.bias_hs + (long long)(q_ok ? qi : 0) * a.bias_ld : nullptr;

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int k_begin = split * a.keys_per_split;
  const int k_end = min(a.Nk, k_begin + a.keys_per_split);

  // staging: thread -> (row, float4 column); D=32: 1 float4 each for K and V, D=64: 2 each
  constexpr int F4_PER_ROW = D / 4;
  constexpr int NLD = 32 * F4_PER_ROW / 256;
  float4 pk[NLD], pv[NLD];
  bool okr[NLD];
  auto gload = [&](int kt) {
    if (WAVES > 4 && tid >= 256) return;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / F4_PER_ROW, c4 = idx % F4_PER_ROW;
      const int key = kt + row;
      okr[i] = key < k_end;                         // branch-free: clamped row, select at LDS-store time
      const int kc = okr[i] ? key : k_begin;
      pk[i] = *reinterpret_cast<const float4*>(kp + (long long)kc * a.k_ld + c4 * 4);
      pv[i] = *reinterpret_cast<const float4*>(vp + (long long)kc * a.v_ld + c4 * 4);
    }
  };
  auto lstore = [&]() {
    if (WAVES > 4 && tid >= 256) return;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / F4_PER_ROW, c4 = idx % F4_PER_ROW;
      const bool ok = okr[i];
      *reinterpret_cast<float4*>(&Ks[row * LDK + c4 * 4]) =
          make_float4(ok ? pk[i].x : 0.f, ok ? pk[i].y : 0.f, ok ? pk[i].z : 0.f, ok ? pk[i].w : 0.f);
      *reinterpret_cast<float4*>(&Vs[row * LDK + c4 * 4]) =
          make_float4(ok ? pv[i].x : 0.f, ok ? pv[i].y : 0.f, ok ? pv[i].z : 0.f, ok ? pv[i].w : 0.f);
    }
  };

  if (k_begin < k_end) gload(k_begin);
  for (int kt = k_begin; kt < k_end; kt += 32) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 32 < k_end) gload(kt + 32);
    if (!wave_active) continue;

    // S^T (keys x queries) = K_tile Q^T
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* krow = &Ks[r32 * LDK + h * HD];
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 kk = *reinterpret_cast<const float4*>(krow + i * 4);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qf[i * 4 + 0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qf[i * 4 + 1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qf[i * 4 + 2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qf[i * 4 + 3], s, 0, 0, 0);
    }
    // mask + running max (scores kept in log2 domain: s * scale * log2(e))
    float mt = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int key0 = kt + 8 * g + 4 * h;   // registers 4g..4g+3 hold keys key0..key0+3
      unsigned mbits = 0;
      if (use_mask) {
        if (key0 + 3 < a.Nk && ((a.mask_ld & 3) == 0)) {
          mbits = *reinterpret_cast<const unsigned*>(mrow + key0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (key0 + e < a.Nk) mbits |= (unsigned)mrow[key0 + e] << (8 * e);
        }
      }
      float bq[4] = {0.f, 0.f, 0.f, 0.f};
      if (brow && key0 < k_end) {                 // bias_ld % 4 == 0 and padded: a float4 never leaves the row
        const float4 bb = *reinterpret_cast<const float4*>(brow + key0);
        bq[0] = bb.x; bq[1] = bb.y; bq[2] = bb.z; bq[3] = bb.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const bool blocked = (key0 + e >= k_end) || ((mbits >> (8 * e)) & 0xffu);
        const float x = blocked ? -INFINITY : (s[r] * a.scale + bq[e]) * 1.4426950408889634f;
        s[r] = x;
        mt = fmaxf(mt, x);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    if (m_new == -INFINITY) {
      // nothing visible yet for this query (wave-divergent per lane is fine: no MFMA skipped below)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
    } else {
      const float alpha = exp2f(m_run - m_new);   // m_run = -inf -> 0
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(s[r] - m_new);
        s[r] = p;
        ps += p;
      }
      ps += __shfl_xor(ps, 32, 64);
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    }
    // O^T (d x queries) += V^T P^T ; step r pairs key (r&3)+8(r>>2) [half 0] with the same +4 [half 1]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const float vv = Vs[key * LDK + t * 32 + r32];
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[r], o[t], 0, 0, 0);
      }
    }
  }

  if (!wave_active || !q_ok) return;
  if (a.nsplit == 1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    const long long obase = b * a.o_bs + (long long)qi * a.o_ld + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * h;
        const float4 r = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
        if (a.out_f16) {
          union { _Float16 hh[4]; uint2 u; } pk;
          pk.hh[0] = (_Float16)r.x; pk.hh[1] = (_Float16)r.y; pk.hh[2] = (_Float16)r.z; pk.hh[3] = (_Float16)r.w;
          *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(a.out) + obase + d) = pk.u;
        } else {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + obase + d) = r;
        }
      }
  } else {
    const long long slot = ((long long)split * a.B * a.H + bh) * a.Nq + qi;
    float* po = a.part_o + slot * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * h;
        *reinterpret_cast<float4*>(po + d) = make_float4(o[t][4 * g], o[t][4 * g + 1], o[t][4 * g + 2], o[t][4 * g + 3]);
      }
    if (h == 0) {
      a.part_ml[slot * 2] = m_run;
      a.part_ml[slot * 2 + 1] = l_run;
    }
  }
}

// merge split-KV partials.  One workgroup of 256 threads per (bh, q): thread = (d = tid % D, split group = tid / D); a group walks
// every G-th split (G = 256 / D groups), the groups' (max, sum, weighted V) are merged through LDS.  (Round 2 ran the nsplit
// -- up to 250 for the 73 600-key level -- partials of a row as ONE serial loop per (bh, q, d) thread: 30 us of dependent loads
// per launch, 9 launches per clip.)
// RAW (the cross-GPU split of ONE clip, ovis_attention_partial_f32): the merged (weighted V, max, sum) of THIS GPU's keys is kept
// un-normalised -- out = packed partial [B*H*Nq*D | B*H*Nq*2 | B*Nq]: the last block holds, per query row, whether any of this GPU's keys
// was open (row_open > 0; rows without an open key were run unmasked by the flash kernel, video decoder:419, and the merge over GPUs
// needs to know which of the two a partial is).
template <int D, bool RAW = false>
__global__ void __launch_bounds__(256)
attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ out,
                    long long o_bs, int o_ld, int B, int H, int Nq, int nsplit, const int* __restrict__ row_open = nullptr,
                    long long open_bs = 0) {
  constexpr int G = 256 / D;
  __shared__ float gm[G], gl[G][D], ga[G][D];
  const int d = threadIdx.x % D, grp = threadIdx.x / D;
  const long long row = blockIdx.x;                       // bh * Nq + q
  const int q = (int)(row % Nq);
  const int bh = (int)(row / Nq);
  const int b = bh / H, head = bh % H;
  const long long BHQ = (long long)B * H * Nq;
  float m = -INFINITY;
  for (int s = grp; s < nsplit; s += G) m = fmaxf(m, part_ml[((long long)s * BHQ + row) * 2]);
  if (d == 0) gm[grp] = m;
  __syncthreads();
  float mall = gm[0];
#pragma unroll
  for (int g = 1; g < G; ++g) mall = fmaxf(mall, gm[g]);
  float l = 0.f, acc = 0.f;
  for (int s = grp; s < nsplit; s += G) {
    const long long slot = (long long)s * BHQ + row;
    const float ms = part_ml[slot * 2];
    const float w = ms == -INFINITY ? 0.f : exp2f(ms - mall);
    l += part_ml[slot * 2 + 1] * w;
    acc += part_o[slot * D + d] * w;
  }
  gl[grp][d] = l; ga[grp][d] = acc;
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int g = 1; g < G; ++g) { l += gl[g][d]; acc += ga[g][d]; }
    if constexpr (RAW) {
      out[row * D + d] = acc;
      if (d == 0) {
        float* ml = out + BHQ * D + row * 2;
        ml[0] = mall; ml[1] = l;
        if (head == 0) out[BHQ * (D + 2) + (long long)b * Nq + q] = (!row_open || row_open[(long long)b * open_bs + q] > 0) ? 1.f : 0.f;
      }
    } else {
      out[b * o_bs + (long long)q * o_ld + head * D + d] = l > 0.f ? acc / l : 0.f;
    }
  }
}

// merge the packed partials of R GPUs (attn_combine_kernel<D, true>, all-gathered: GPU r's block at parts + r * stride) into the attention
// output.  One 64-thread workgroup per (bh, q); R is the number of GPUs of one node (<= 8 here), a serial loop.  A query row that has an
// open key on ANY GPU takes only the partials of the GPUs where it has one (the others ran the row unmasked); a row closed everywhere takes
// all of them -- together exactly "rows whose mask blocks every key of the clip attend to everything" (video decoder:419).
template <int D>
__global__ void __launch_bounds__(64)
attn_merge_ranks_kernel(const float* __restrict__ parts, int R, long long stride, float* __restrict__ out, long long o_bs, int o_ld,
                        int B, int H, int Nq) {
  const int d = threadIdx.x;
  if (d >= D) return;
  const long long row = blockIdx.x;
  const int q = (int)(row % Nq);
  const int bh = (int)(row / Nq);
  const int b = bh / H, head = bh % H;
  const long long BHQ = (long long)B * H * Nq;
  const long long flag_at = BHQ * (D + 2) + (long long)b * Nq + q;
  bool any_open = false;
  for (int r = 0; r < R; ++r) any_open |= parts[r * stride + flag_at] > 0.f;
  float m = -INFINITY;
  for (int r = 0; r < R; ++r) {
    const float* p = parts + r * stride;
    if (any_open && !(p[flag_at] > 0.f)) continue;
    m = fmaxf(m, p[BHQ * D + row * 2]);
  }
  float l = 0.f, acc = 0.f;
  for (int r = 0; r < R; ++r) {
    const float* p = parts + r * stride;
    if (any_open && !(p[flag_at] > 0.f)) continue;
    const float mr = p[BHQ * D + row * 2];
    const float w = mr == -INFINITY ? 0.f : exp2f(mr - m);
    l += p[BHQ * D + row * 2 + 1] * w;
    acc += p[row * D + d] * w;
  }
  out[b * o_bs + (long long)q * o_ld + head * D + d] = l > 0.f ? acc / l : 0.f;
}

}  // namespace

extern "C" long long ovis_attention_workspace_bytes(int B, int H, int Nq, int D, int nsplit) {
  if (nsplit <= 1) return 0;
  return (long long)nsplit * B * H * Nq * (D + 2) * sizeof(float);
}

// `partial` != null: the cross-GPU form -- the flash kernel always writes split partials (AttnArgs.nsplit = 0 stands for "one split, kept
// raw": the kernel normalises in place only when the field is 1) and the RAW combine packs this GPU's merged partial into `partial`.
static int attention_launch(const float* q, long long q_bs, int q_ld, const float* k, long long k_bs, int k_ld,
                            const float* v, long long v_bs, int v_ld, void* out, long long o_bs, int o_ld,
                            int out_f16, const uint8_t* mask, long long mask_ld, long long mask_bs,
                            const int* row_open, const float* bias, long long bias_bs, long long bias_hs,
                            int bias_ld, int B, int H, int Nq, int Nk, int D, float scale, int nsplit,
                            float* workspace, float* partial, ovis_stream_t stream) {
  if (partial) out = partial;
  OVIS_REQUIRE(q && k && v && out, "attention: null pointer");
  OVIS_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attention: non-positive size");
  OVIS_REQUIRE(D == 32 || D == 64, "attention: head dim %d not supported (32 or 64)", D);
  OVIS_REQUIRE(q_ld % 4 == 0 && k_ld % 4 == 0 && v_ld % 4 == 0 && o_ld % 4 == 0 && q_bs % 4 == 0 && k_bs % 4 == 0 &&
                   v_bs % 4 == 0 && o_bs % 4 == 0,
               "attention: strides must be multiples of 4 floats");
  OVIS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "attention: 16-byte alignment");
  OVIS_REQUIRE(nsplit >= 1 && ((nsplit == 1 && !partial) || workspace), "attention: nsplit > 1 (and every partial launch) needs a workspace");
  OVIS_REQUIRE(!(out_f16 && nsplit > 1), "attention: fp16 output is only supported with nsplit == 1");
  OVIS_REQUIRE(!mask || mask_ld >= Nk, "attention: mask_ld < Nk");
  OVIS_REQUIRE(!bias || (bias_ld % 4 == 0 && bias_ld >= (Nk + 3) / 4 * 4 && bias_bs % 4 == 0 && bias_hs % 4 == 0 &&
                         ((uintptr_t)bias & 15) == 0),
               "attention: bias rows must be padded to a multiple of 4 floats and 16-byte aligned");
  int keys_per_split = (Nk + nsplit - 1) / nsplit;
  keys_per_split = (keys_per_split + 31) / 32 * 32;
  nsplit = (Nk + keys_per_split - 1) / keys_per_split;
  AttnArgs a;
  a.q = q; a.q_bs = q_bs; a.q_ld = q_ld; a.k = k; a.k_bs = k_bs; a.k_ld = k_ld; a.v = v; a.v_bs = v_bs; a.v_ld = v_ld;
  a.out = out; a.o_bs = o_bs; a.o_ld = o_ld; a.out_f16 = out_f16; a.mask = mask; a.mask_ld = mask_ld; a.mask_bs = mask_bs; a.row_open = row_open;
  a.bias = bias; a.bias_bs = bias_bs; a.bias_hs = bias_hs; a.bias_ld = bias_ld;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.nsplit = (partial && nsplit == 1) ? 0 : nsplit; a.keys_per_split = keys_per_split; a.scale = scale;
  a.part_o = workspace;
  a.part_ml = workspace ? workspace + (long long)nsplit * B * H * Nq * D : nullptr;
  const bool five = Nq > 128 && Nq <= 160;                          // one 160-query workgroup instead of 128 + a nearly empty one
  dim3 grid(five ? 1 : ovis::cdiv(Nq, 128), B * H, nsplit);
  hipStream_t s = (hipStream_t)stream;
  if (five) {
    if (D == 32) hipLaunchKernelGGL((flash_attn_f32_kernel<32, 5>), grid, dim3(320), 0, s, a);
    else hipLaunchKernelGGL((flash_attn_f32_kernel<64, 5>), grid, dim3(320), 0, s, a);
  } else if (D == 32) hipLaunchKernelGGL(flash_attn_f32_kernel<32>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(flash_attn_f32_kernel<64>, grid, dim3(256), 0, s, a);
  int rc = ovis::check_launch("attention");
  if (rc) return rc;
  const unsigned rows = (unsigned)((long long)B * H * Nq);
  if (partial) {
    const long long open_bs = mask_bs ? Nq : 0;
    if (D == 32) hipLaunchKernelGGL((attn_combine_kernel<32, true>), dim3(rows), dim3(256), 0, s, a.part_o, a.part_ml, partial, 0LL, 0, B, H, Nq, nsplit, row_open, open_bs);
    else hipLaunchKernelGGL((attn_combine_kernel<64, true>), dim3(rows), dim3(256), 0, s, a.part_o, a.part_ml, partial, 0LL, 0, B, H, Nq, nsplit, row_open, open_bs);
    rc = ovis::check_launch("attention combine (partial)");
  } else if (nsplit > 1) {
    if (D == 32) hipLaunchKernelGGL(attn_combine_kernel<32>, dim3(rows), dim3(256), 0, s, a.part_o, a.part_ml, (float*)out, o_bs, o_ld, B, H, Nq, nsplit);
    else hipLaunchKernelGGL(attn_combine_kernel<64>, dim3(rows), dim3(256), 0, s, a.part_o, a.part_ml, (float*)out, o_bs, o_ld, B, H, Nq, nsplit);
    rc = ovis::check_launch("attention combine");
  }
  return rc;
}

extern "C" int ovis_attention_f32(const float* q, long long q_bs, int q_ld, const float* k, long long k_bs, int k_ld,
                                  const float* v, long long v_bs, int v_ld, void* out, long long o_bs, int o_ld,
                                  int out_f16, const uint8_t* mask, long long mask_ld, long long mask_bs,
                                  const int* row_open, const float* bias, long long bias_bs, long long bias_hs,
                                  int bias_ld, int B, int H, int Nq, int Nk, int D, float scale, int nsplit,
                                  float* workspace, ovis_stream_t stream) {
  return attention_launch(q, q_bs, q_ld, k, k_bs, k_ld, v, v_bs, v_ld, out, o_bs, o_ld, out_f16, mask, mask_ld, mask_bs, row_open, bias,
                          bias_bs, bias_hs, bias_ld, B, H, Nq, Nk, D, scale, nsplit, workspace, nullptr, stream);
}

extern "C" long long ovis_attention_partial_floats(int B, int H, int Nq, int D) {
  return (long long)B * H * Nq * (D + 2) + (long long)B * Nq;
}

extern "C" long long ovis_attention_partial_workspace_bytes(int B, int H, int Nq, int D, int nsplit) {
  return (long long)(nsplit < 1 ? 1 : nsplit) * B * H * Nq * (D + 2) * sizeof(float);
}

extern "C" int ovis_attention_partial_f32(const float* q, long long q_bs, int q_ld, const float* k, long long k_bs, int k_ld,
                                          const float* v, long long v_bs, int v_ld, const uint8_t* mask, long long mask_ld,
                                          long long mask_bs, const int* row_open, int B, int H, int Nq, int Nk, int D, float scale,
                                          int nsplit, float* workspace, float* partial, ovis_stream_t stream) {
  OVIS_REQUIRE(partial && workspace, "attention partial: null pointer");
  OVIS_REQUIRE((((uintptr_t)partial) & 15) == 0, "attention partial: 16-byte alignment");
  return attention_launch(q, q_bs, q_ld, k, k_bs, k_ld, v, v_bs, v_ld, nullptr, 4, 4, 0, mask, mask_ld, mask_bs, row_open, nullptr, 0, 0, 0,
                          B, H, Nq, Nk, D, scale, nsplit, workspace, partial, stream);
}

extern "C" int ovis_attention_merge_f32(const float* parts, int R, long long stride, float* out, long long o_bs, int o_ld, int B, int H,
                                        int Nq, int D, ovis_stream_t stream) {
  OVIS_REQUIRE(parts && out, "attention merge: null pointer");
  OVIS_REQUIRE(R >= 1 && B > 0 && H > 0 && Nq > 0, "attention merge: non-positive size");
  OVIS_REQUIRE(D == 32 || D == 64, "attention merge: head dim %d not supported (32 or 64)", D);
  OVIS_REQUIRE(stride >= (long long)B * H * Nq * (D + 2) + (long long)B * Nq, "attention merge: stride smaller than one partial");
  const unsigned rows = (unsigned)((long long)B * H * Nq);
  hipStream_t s = (hipStream_t)stream;
  if (D == 32) hipLaunchKernelGGL(attn_merge_ranks_kernel<32>, dim3(rows), dim3(64), 0, s, parts, R, stride, out, o_bs, o_ld, B, H, Nq);
  else hipLaunchKernelGGL(attn_merge_ranks_kernel<64>, dim3(rows), dim3(64), 0, s, parts, R, stride, out, o_bs, o_ld, B, H, Nq);
  return ovis::check_launch("attention merge");
}
