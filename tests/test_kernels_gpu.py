"""GPU: individual HIP kernels vs plain torch references of the same op (fp64/fp32 on CPU)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(out, ref):
    return (out.double().cpu() - ref.double()).abs().max().item() / (ref.double().abs().max().item() + 1e-12)


@pytest.mark.parametrize("M,N,K", [(197 * 3, 768, 768), (1000, 3072, 768), (130, 96, 64), (4000, 2304, 768),
                                   (16389, 4100, 256), (40000, 768, 3072)])     # the last two take the 256x256 ring kernel
@pytest.mark.parametrize("out_f16", [False, True])
def test_gemm_f16(M, N, K, out_f16):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().T + b.double() + r.double()
    ref = ref * torch.sigmoid(1.702 * ref)
    out = ops.gemm_nt_f16(a.cuda(), w.cuda(), b.cuda(), r.cuda(), ops.ACT_QUICKGELU, out_f16=out_f16)
    assert out.dtype == (torch.float16 if out_f16 else torch.float32)
    assert _rel(out, ref) < (2e-3 if out_f16 else 2e-5)


@pytest.mark.parametrize("M,N,K", [(150, 70, 128), (2001, 1027, 128), (4099, 768, 192), (8200, 2060, 128),
                                   (16384, 4096, 64), (25500, 768, 128)])       # small-tile, LDS-DMA 128 and 256-ring paths;
                                                                                # (8200,2060) and (25500,768) split off a tail round
def test_gemm_f16_integer_exact(M, N, K):
    from openvis_amd import ops
    a = (torch.arange(M * K).reshape(M, K) % 13 - 6).half()
    w = (torch.arange(N * K).reshape(N, K) % 7 - 3).half()
    out = ops.gemm_nt_f16(a.cuda(), w.cuda())
    ref = (a.cuda().float() @ w.cuda().float().T)      # exact in f32 for these small integers
    assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K", [(100, 256, 256), (96600 // 5, 256, 256), (40000, 512, 2048), (777, 130, 264)])
def test_gemm_f32a_f16w(M, N, K):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = (a.half().double() @ w.half().double().T + b.double() + r.double()).relu()    # fp16-rounded operands, exact sum
    wd = w.cuda()
    out = ops.gemm_nt(a.cuda(), wd, b.cuda(), r.cuda(), ops.ACT_RELU, w16=ops.cast_f16(wd))
    assert _rel(out, ref) < 2e-5


def test_conv_f32a_f16w():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(2)
    for (N, H, W, Cin, Cout, k, s_, p_) in [(2, 23, 40, 64, 64, 3, 1, 1), (1, 46, 80, 128, 256, 3, 2, 1), (2, 12, 20, 256, 512, 1, 2, 0)]:
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x.half().double(), w.half().double(), b.double(), stride=s_, padding=p_).relu()
        wd = w.permute(0, 2, 3, 1).contiguous().cuda()
        y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wd, s_, p_, b.cuda(), None, ops.ACT_RELU, w16=ops.cast_f16(wd))
        assert _rel(y.permute(0, 3, 1, 2), ref) < 2e-5


def _attn_ref(q, k, v, mask=None):
    # q [B,Nq,H,D] ...; mask bool [Nq,Nk] True = blocked (rows fully blocked -> unmasked)
    s = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) / math.sqrt(q.shape[-1])
    if mask is not None:
        m = mask.clone()
        m[m.all(-1)] = False
        s = s.masked_fill(m[None, None], float("-inf"))
    return torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.double()).flatten(2)


@pytest.mark.parametrize("B,H,Nq,Nk,D,nsplit,masked", [
    (1, 8, 100, 4600, 32, 4, True), (1, 8, 100, 100, 32, 1, False), (3, 12, 197, 197, 64, 1, False),
    (1, 8, 100, 1000, 32, 1, True), (2, 4, 37, 333, 64, 3, False), (1, 8, 100, 73600 // 8, 32, 8, True),
    (7, 6, 144, 144, 32, 1, False), (2, 3, 160, 300, 64, 2, False), (1, 4, 129, 129, 32, 1, True)])   # 128 < Nq <= 160: the 5-wavefront variant
def test_attention(B, H, Nq, Nk, D, nsplit, masked):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(Nq * Nk)
    C = H * D
    q = torch.randn(B, Nq, C, generator=g)
    kv = torch.randn(B, Nk, 2 * C, generator=g)
    mask = None
    if masked:
        logits = torch.randn(Nq, Nk, generator=g) + 1.0
        logits[3] = -5.0           # a fully blocked row -> must be treated as unmasked
        logits[7, : Nk // 2] = -5.0
        mask = torch.sigmoid(logits) < 0.5
    ref = _attn_ref(q.view(B, Nq, H, D), kv[..., :C].reshape(B, Nk, H, D), kv[..., C:].reshape(B, Nk, H, D), mask)
    qd, kvd = q.cuda(), kv.cuda()
    md = ro = None
    if masked:
        md, ro = ops.attn_mask_from_logits(logits.cuda())
        assert torch.equal(md[:, :Nk].cpu().bool(), mask)
        assert torch.equal(ro.cpu(), (~mask).sum(-1).int())
    out = ops.attention(qd, kvd, kvd[..., C:], B, H, Nq, Nk, D, Nq * C, C, Nk * 2 * C, 2 * C, Nk * 2 * C, 2 * C, md, ro, nsplit)
    assert _rel(out, ref) < 2e-5
    if nsplit == 1:
        out16 = ops.attention(qd, kvd, kvd[..., C:], B, H, Nq, Nk, D, Nq * C, C, Nk * 2 * C, 2 * C, Nk * 2 * C, 2 * C, md, ro, 1,
                              out_f16=True)
        assert out16.dtype == torch.float16 and _rel(out16, ref) < 2e-3


@pytest.mark.parametrize("H,Nq,D,blocks,nsplit", [(8, 100, 32, (2300, 1500), 8), (8, 100, 32, (920, 920, 920, 460), 3), (4, 37, 64, (65, 31, 200), 1),
                                                  (8, 100, 32, (18400,), 64)])
def test_attention_keys_split_over_gpus(H, Nq, D, blocks, nsplit):
    """ovis_attention_partial_f32 + ovis_attention_merge_f32 (the split-KV form of the offline video decoder): the key range cut
    into contiguous blocks as the frames of a clip are over GPUs; each block's packed partial, stacked as an all-gather would, merged.
    Rows: 3 is blocked on EVERY key (attends to all), 7 is blocked on the whole first block only (that block's partial must be dropped),
    11 is open on the first block only."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(Nq + sum(blocks))
    C, Nk = H * D, sum(blocks)
    q = torch.randn(1, Nq, C, generator=g)
    kv = torch.randn(1, Nk, 2 * C, generator=g)
    logits = torch.randn(Nq, Nk, generator=g) + 1.0
    logits[3] = -5.0
    logits[7, : blocks[0]] = -5.0
    logits[11, blocks[0]:] = -5.0
    mask = torch.sigmoid(logits) < 0.5
    ref = _attn_ref(q.view(1, Nq, H, D), kv[..., :C].reshape(1, Nk, H, D), kv[..., C:].reshape(1, Nk, H, D), mask)
    qd = q.cuda()
    parts, k0 = [], 0
    for n in blocks:
        kvb = kv[:, k0:k0 + n].contiguous().cuda()
        md, ro = ops.attn_mask_from_logits(logits[:, k0:k0 + n].contiguous().cuda())
        parts.append(ops.attention_partial(qd, kvb, kvb[..., C:], 1, H, Nq, n, D, Nq * C, C, n * 2 * C, 2 * C, n * 2 * C, 2 * C, md, ro, nsplit))
        k0 += n
    out = ops.attention_merge(torch.stack(parts).contiguous(), 1, H, Nq, D)
    assert _rel(out, ref) < 2e-5
    if len(blocks) == 1:            # one GPU: the same partials merged by the same arithmetic as ovis_attention_f32 -- bit for bit
        kvd = kv.cuda()
        md, ro = ops.attn_mask_from_logits(logits.cuda())
        whole = ops.attention(qd, kvd, kvd[..., C:], 1, H, Nq, Nk, D, Nq * C, C, Nk * 2 * C, 2 * C, Nk * 2 * C, 2 * C, md, ro, nsplit)
        assert torch.equal(out, whole)
    # without a mask (row_open NULL): every block counts as open
    parts, k0 = [], 0
    for n in blocks:
        kvb = kv[:, k0:k0 + n].contiguous().cuda()
        parts.append(ops.attention_partial(qd, kvb, kvb[..., C:], 1, H, Nq, n, D, Nq * C, C, n * 2 * C, 2 * C, n * 2 * C, 2 * C, None, None, nsplit))
        k0 += n
    ref0 = _attn_ref(q.view(1, Nq, H, D), kv[..., :C].reshape(1, Nk, H, D), kv[..., C:].reshape(1, Nk, H, D))
    assert _rel(ops.attention_merge(torch.stack(parts).contiguous(), 1, H, Nq, D), ref0) < 2e-5


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_attention_keys_split_per_batch_masks_and_random_blocks(seed):
    """The same pair of entry points with B > 1 and one mask per batch element (mask_bs != 0: the frame decoders' form,
    frame_mask2former_transformer_decoder.py:85-94), key blocks of random sizes down to ONE key, and rows closed in random patterns."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(1000 + seed)
    B, H, Nq, D = 3, 8, 100, 32
    C = H * D
    blocks = [int(v) for v in torch.randint(1, 700, (int(torch.randint(2, 7, (1,), generator=g)),), generator=g)] + [1]
    Nk = sum(blocks)
    q = torch.randn(B, Nq, C, generator=g)
    kv = torch.randn(B, Nk, 2 * C, generator=g)
    logits = torch.randn(B * Nq, Nk, generator=g) + 0.5
    edges = [0] + list(torch.tensor(blocks).cumsum(0))
    for r in torch.randint(0, B * Nq, (40,), generator=g).tolist():           # close rows on random runs of blocks (some on all of them)
        i, j = sorted(torch.randint(0, len(blocks) + 1, (2,), generator=g).tolist())
        logits[r, int(edges[i]):int(edges[j])] = -6.0
    logits[5] = -6.0
    mask = (torch.sigmoid(logits) < 0.5).view(B, Nq, Nk)
    ref = torch.cat([_attn_ref(q[b:b + 1].view(1, Nq, H, D), kv[b:b + 1, :, :C].reshape(1, Nk, H, D), kv[b:b + 1, :, C:].reshape(1, Nk, H, D), mask[b])
                     for b in range(B)])
    qd = q.cuda()
    parts = []
    for i, n in enumerate(blocks):
        k0 = int(edges[i])
        kvb = kv[:, k0:k0 + n].contiguous().cuda()
        md, ro = ops.attn_mask_from_logits(logits[:, k0:k0 + n].contiguous().cuda())
        parts.append(ops.attention_partial(qd, kvb, kvb[..., C:], B, H, Nq, n, D, Nq * C, C, n * 2 * C, 2 * C, n * 2 * C, 2 * C, md, ro,
                                           nsplit=1 + i % 3, mask_per_batch=True))
    out = ops.attention_merge(torch.stack(parts).contiguous(), B, H, Nq, D)
    assert _rel(out, ref) < 2e-5


@pytest.mark.parametrize("B,H,N", [(3, 12, 197), (2, 4, 64), (1, 16, 577), (5, 2, 33), (2, 3, 193), (2, 3, 205), (2, 3, 215), (2, 3, 224), (2, 2, 208)])
def test_attention_f16(B, H, N):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(N)
    D, C = 64, H * 64
    qkv = (torch.randn(B * N, 3 * C, generator=g)).half()
    q, k, v = [qkv[:, i * C:(i + 1) * C].float().reshape(B, N, H, D) for i in range(3)]
    ref = _attn_ref(q, k, v)
    qd = qkv.cuda()
    out = ops.attention_f16(qd, qd[:, C:], qd[:, 2 * C:], B, H, N, N, D, N * 3 * C, 3 * C, N * 3 * C, 3 * C, N * 3 * C, 3 * C)
    assert out.dtype == torch.float16 and _rel(out, ref) < 3e-3
    # exact-integer layout check: one-hot attention (huge logit on key (7q+3) % N) must copy that V row exactly
    vi = torch.randint(-8, 9, (B, N, H, D)).float()
    idx = (7 * torch.arange(N) + 3) % N
    # keys: make key idx[q] the (near-)unique maximiser for query q via a +-1 code
    code = torch.randn(N, D, generator=g).sign()
    qi = code[None, :, None, :].expand(B, N, H, D).clone() * 4.0
    ki = torch.zeros(B, N, H, D)
    ki[:, idx] = qi                                                # key idx[q] == query q's code -> score 16*D/8 >> others
    x = torch.cat([qi.reshape(B * N, C), ki.reshape(B * N, C), vi.reshape(B * N, C)], 1).half().cuda()
    out = ops.attention_f16(x, x[:, C:], x[:, 2 * C:], B, H, N, N, D, N * 3 * C, 3 * C, N * 3 * C, 3 * C, N * 3 * C, 3 * C)
    ref = _attn_ref(qi, ki, vi)
    assert _rel(out, ref) < 3e-3


def test_layernorm_groupnorm_maxpool():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(5)
    for C in (256, 768, 2048):
        x = torch.randn(333, C, generator=g) * 3 + 1
        r = torch.randn(333, C, generator=g)
        w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
        ref = F.layer_norm((x + r).double(), (C,), w.double(), b.double())
        assert _rel(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), r.cuda()), ref) < 1e-5
        assert _rel(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), r.cuda(), out_f16=True), ref) < 2e-3
    x = torch.randn(2, 256, 23, 40, generator=g) * 2 + 0.5
    up = torch.randn(2, 256, 12, 20, generator=g)
    w, b = torch.randn(256, generator=g), torch.randn(256, generator=g)
    ref = F.group_norm(x.double(), 32, w.double(), b.double()) + F.interpolate(up.double(), size=(23, 40), mode="bilinear",
                                                                                align_corners=False)
    y = ops.groupnorm_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), b.cuda(),
                           up_add=up.permute(0, 2, 3, 1).contiguous().cuda())
    assert _rel(y.permute(0, 3, 1, 2), ref) < 1e-5
    y = ops.groupnorm_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda(), b.cuda(), relu=True)
    assert _rel(y.permute(0, 3, 1, 2), F.group_norm(x.double(), 32, w.double(), b.double()).relu()) < 1e-5
    x = torch.randn(2, 64, 37, 51, generator=g)
    y = ops.maxpool3x3s2(x.permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 3, 2, 1))


def test_position_encodings_and_pool():
    from openvis_amd import ops
    from oracle import torch_ref as TR
    pe2 = ops.pe_sine(1, 23, 40, 128, False, None, "cuda").cpu()
    assert (pe2[0].permute(2, 0, 1) - TR.pe_sine_2d(1, 23, 40)[0]).abs().max() < 2e-5
    pe3 = ops.pe_sine(5, 12, 20, 128, True, None, "cuda").cpu()
    assert (pe3.permute(0, 3, 1, 2) - TR.pe_sine_3d(1, 5, 12, 20)[0]).abs().max() < 2e-5
    x = torch.randn(2, 16, 24, 8)
    for s in (2, 4, 8):
        ref = F.interpolate(x.permute(0, 3, 1, 2), size=(16 // s, 24 // s), mode="bilinear", align_corners=False)
        assert (ops.center_pool(x.cuda(), s).permute(0, 3, 1, 2).cpu() - ref).abs().max() < 1e-6


def test_msda_encoder_fused_matches_unfused_oracle():
    from openvis_amd import ops
    from oracle import torch_ref as TR
    g = torch.Generator().manual_seed(11)
    sizes = [(4, 7), (8, 14), (15, 27)]
    shapes = torch.tensor(sizes)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    B = 2
    value = torch.randn(B, S, 256, generator=g)
    oa = torch.randn(B, S, 288, generator=g)
    oa[..., :192] *= 3.0
    off = oa[..., :192].reshape(B, S, 8, 3, 4, 2)
    aw = oa[..., 192:].reshape(B, S, 8, 12).softmax(-1).view(B, S, 8, 3, 4)
    ref_pts = TR.encoder_reference_points(sizes).expand(B, -1, -1, -1)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    ref = TR.msda_torch(value.view(B, S, 8, 32).double(), shapes, lsi, loc.double(), aw.double())
    out = ops.msda_encoder_fused(value.cuda(), oa.cuda(), shapes.cuda(), lsi.cuda())
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize("sizes,B,scale", [([(4, 7), (8, 14), (15, 27)], 2, 3.0), ([(3, 5), (6, 10), (11, 19)], 3, 40.0),
                                           ([(23, 40), (46, 80), (92, 160)], 5, 2.0), ([(1, 1), (2, 2), (3, 5)], 1, 1.0)])
def test_msda_lane_sharing_kernel_is_bit_identical(sizes, B, scale):
    """msda_encoder_fused8_kernel (one lane of a (query, head) group computes a sampling point, the other 7 fetch it with ds_swizzle)
    against msda_encoder_fused_kernel (every lane computes every point): same expressions, same order -> identical bits.  Cases: ragged
    item counts (not a multiple of the 256-thread block), offsets far outside the maps (scale 40: border rules), the full 720p x 5
    shape of the bench, a 1x1 level."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(len(sizes) * 100 + B)
    shapes = torch.tensor(sizes)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, 256, generator=g).cuda()
    oa = torch.randn(B, S, 288, generator=g)
    oa[..., :192] *= scale
    oa = oa.cuda()
    try:
        ops.msda_set_share(False)
        ref = ops.msda_encoder_fused(value, oa, shapes.cuda(), lsi.cuda())
        ops.msda_set_share(True)
        out = ops.msda_encoder_fused(value, oa, shapes.cuda(), lsi.cuda())
    finally:
        ops.msda_set_share(True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert out.abs().max().item() > 0


@pytest.mark.parametrize("Q,T,h,w,kind", [(7, 2, 12, 16, "noise"), (5, 3, 23, 40, "blobs"), (4, 1, 184, 320, "noise"), (6, 2, 46, 80, "tiny"),
                                          (3, 2, 5, 3, "noise"), (4, 2, 1, 1, "noise")])
def test_mask_bbox_cell_kernel_gives_identical_boxes(Q, T, h, w, kind):
    """mask_bbox4_kernel (one thread per low-resolution pixel: 3x3 neighbourhood, 16 output pixels, whole-cell shortcuts) against the
    per-pixel mask_bbox_kernel: identical boxes on noise around the threshold, smooth blobs, values within 1e-5 of zero, empty and
    full masks, and maps narrower than the 3x3 neighbourhood."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(Q * 1000 + h)
    if kind == "noise":
        m = torch.randn(Q, T, h, w, generator=g)
    elif kind == "tiny":
        m = (torch.rand(Q, T, h, w, generator=g) - 0.5) * 2e-5                # |x| <= 1e-5: the sigmoid branch of mask_on
    else:
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        c = torch.rand(Q, T, 2, generator=g)
        m = 3.0 - ((yy[None, None] - c[..., 0, None, None] * h) ** 2 + (xx[None, None] - c[..., 1, None, None] * w) ** 2) / (0.02 * h * w + 1)
    m[0] = -2.0                                                                # empty
    m[-1, -1] = 5.0                                                            # full
    m = m.contiguous().cuda()
    try:
        ops.mask_bbox_set_cells(False)
        ref = ops.mask_bbox(m, 4 * h, 4 * w)
        ops.mask_bbox_set_cells(True)
        out = ops.mask_bbox(m, 4 * h, 4 * w)
    finally:
        ops.mask_bbox_set_cells(True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), (out.cpu() - ref.cpu()).abs().max()
    assert int(ref[:, 0, 2].max()) < 0 and ref[-1, -1].tolist() == [0, 0, 4 * w - 1, 4 * h - 1]


def test_mask_bbox_crop_and_final_masks_vs_oracle():
    from openvis_amd import ops
    from oracle import torch_ref as TR
    g = torch.Generator().manual_seed(3)
    Q, T, h, w, H, W = 6, 2, 12, 16, 45, 60
    Hp, Wp = 48, 64
    masks = torch.randn(Q, T, h, w, generator=g) * 2 - 1.0
    masks[1] = -3.0                                              # never valid
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8)
    up = F.interpolate(masks, size=(Hp, Wp), mode="bilinear", align_corners=False)
    part = up.sigmoid().transpose(0, 1).contiguous()
    regions, valid, sboxes = TR.clip_crops(frames, part, 32)
    boxes = ops.mask_bbox(masks.cuda(), Hp, Wp).cpu().numpy()
    assert np.array_equal(boxes[..., 2] >= 0, valid.numpy())
    tq = np.argwhere(valid.numpy())
    crops = np.concatenate([tq, boxes[valid.numpy()]], 1).astype(np.int32)
    side = np.maximum(crops[:, 4] + 1 - crops[:, 2], crops[:, 5] + 1 - crops[:, 3])
    assert np.array_equal(np.stack([crops[:, 2], crops[:, 3], crops[:, 2] + side, crops[:, 3] + side], 1), sboxes.numpy())
    A = ops.clip_crop_patches(frames.cuda(), masks.cuda(), torch.from_numpy(crops).cuda(), Hp, Wp, 32, 16, TR.CLIP_MEAN,
                              TR.CLIP_STD).cpu()
    mean = torch.tensor(TR.CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(TR.CLIP_STD).view(1, 3, 1, 1)
    img = (regions / 255. - mean) / std                         # [M,3,32,32]
    ref_A = img.unfold(2, 16, 16).unfold(3, 16, 16).permute(0, 2, 3, 1, 4, 5).reshape(-1, 3 * 256)
    assert (A - ref_A).abs().max() < 2e-4
    sel = torch.tensor([0, 2, 5], dtype=torch.int32)
    for (OH, OW) in ((H, W), (90, 120)):
        out = ops.final_masks(masks.cuda(), sel.cuda(), Hp, Wp, H, W, OH, OW).cpu().bool()
        ref = F.interpolate(up[sel.long()][:, :, :H, :W], size=(OH, OW), mode="bilinear", align_corners=False) > 0
        assert (out == ref).float().mean() > 0.9995


@pytest.mark.parametrize("H,W,cm", [(720, 1280, False), (720, 1280, True), (90, 120, False), (88, 118, True), (45, 60, False), (3, 4, False), (4, 3, True)])
def test_final_masks_cell_kernel_is_bit_identical_to_the_per_pixel_kernel(H, W, cm):
    """final_masks_cell_kernel (one thread per low-resolution cell: 3 x 3 neighbourhood -> a 4 x 4 block of output pixels by the same
    make_tap / bilerp expressions) against final_masks_kernel<1>: output == image size, exact x4 padded maps, ragged last block rows /
    columns (H or W not a multiple of 4 on the slow axis), clamped borders, row- and column-major."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(H * 7 + W)
    Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    m = (torch.randn(6, 2, Hp // 4, Wp // 4, generator=g) * (1e-3 if H == 90 else 2.0)).cuda()     # one case with logits near the threshold
    sel = torch.tensor([5, 0, 3], dtype=torch.int32).cuda()
    try:
        ops.final_masks_set_cells(False)
        ref = ops.final_masks(m, sel, Hp, Wp, H, W, H, W, column_major=cm)
        ops.final_masks_set_cells(True)
        out = ops.final_masks(m, sel, Hp, Wp, H, W, H, W, column_major=cm)
    finally:
        ops.final_masks_set_cells(True)
    torch.cuda.synchronize()
    assert out.shape == ref.shape and torch.equal(out, ref)
    assert 0 < int(ref.sum()) < ref.numel()


def test_aggregate_topk():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(9)
    T, Q, K = 3, 10, 17
    valid = torch.rand(T, Q, generator=g) > 0.4
    valid[:, 4] = False
    M = int(valid.sum())
    logits = torch.randn(M, K, generator=g) * 2
    slot = -torch.ones(T, Q, dtype=torch.int32)
    slot[valid] = torch.arange(M, dtype=torch.int32)
    probs, qv = ops.openvis_aggregate(logits.cuda(), slot.cuda())
    ids = torch.nonzero(valid)
    vq = torch.nonzero(valid.any(0))[:, 0]
    ref = torch.stack([logits[ids[:, 1] == q].mean(0) for q in vq]).softmax(-1)
    assert torch.equal(qv.cpu().bool(), valid.any(0))
    assert (probs.cpu()[vq] - ref).abs().max() < 1e-6
    idx, score, ent, sel_q = ops.topk_entropy(probs, vq.int().cuda(), 10)
    assert torch.equal(sel_q.cpu().long(), vq[(idx.cpu().long() // probs.shape[1])].long())
    rs, ri = ref.flatten().topk(10)
    assert set(idx.cpu().tolist()) == set(ri.tolist())
    ent_ref = {int(i): float((-ref[i // K] * ref[i // K].log()).sum()) for i in ri}
    for i, e in zip(idx.cpu().tolist(), ent.cpu().tolist()):
        assert abs(e - ent_ref[i]) < 1e-5


@pytest.mark.parametrize("nrows,Q,K,topk", [(100, 100, 482, 10), (37, 200, 1203, 10), (3, 5, 2000, 64), (1, 4, 7, 7)])
def test_topk_entropy_sizes_order_and_ties(nrows, Q, K, topk):
    """flat top-k over the valid rows == torch.topk on the gathered rows (values in descending order, ties to the smaller
    flat index), query ids, entropies; K below / above the workgroup size, topk up to the kernel's 64."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(nrows * K)
    probs = torch.rand(Q, K, generator=g).softmax(-1)
    rows = torch.randperm(Q, generator=g)[:nrows].sort().values
    probs[rows[0], 3 % K] = probs[rows[-1], 5 % K] = 0.5            # an exact tie at the top
    idx, score, ent, sel_q = ops.topk_entropy(probs.cuda(), rows.int().cuda(), topk)
    flat = probs[rows].flatten()
    order = sorted(range(flat.numel()), key=lambda i: (-float(flat[i]), i))[:topk]
    assert idx.cpu().tolist() == order
    assert torch.equal(score.cpu(), flat[order])
    assert sel_q.cpu().tolist() == [int(rows[i // K]) for i in order]
    ref_e = torch.stack([(-probs[rows[i // K]] * probs[rows[i // K]].log()).sum() for i in order])
    assert (ent.cpu() - ref_e).abs().max() < 1e-5


def test_topk_entropy_randomised_with_heavy_ties():
    """scores quantised to a few levels (ties everywhere: the order is decided by the flat index), random sizes incl. more
    than 128 elements per thread (beyond the kernel's own-selection bitmask) and threads that win several times"""
    from openvis_amd import ops
    rng = np.random.default_rng(11)
    for case in range(24):
        Q = int(rng.integers(1, 260))
        K = int(rng.integers(1, 1300))
        nrows = int(rng.integers(1, Q + 1))
        topk = int(min(rng.integers(1, 65), nrows * K))
        g = torch.Generator().manual_seed(case)
        levels = int(rng.integers(2, 50))
        probs = (torch.randint(1, levels + 1, (Q, K), generator=g).float() / (levels + 1)).contiguous()
        if case % 3 == 0:                                    # one hot row: the same threads win again and again
            probs[int(rng.integers(0, Q))] = 0.999
        rows = torch.randperm(Q, generator=g)[:nrows].sort().values
        idx, score, ent, sel_q = ops.topk_entropy(probs.cuda(), rows.int().cuda(), topk)
        flat = probs[rows].flatten()
        # descending value, ascending index on ties
        order = sorted(range(flat.numel()), key=lambda i: (-float(flat[i]), i))[:topk] if flat.numel() < 20000 else None
        if order is None:
            v, _ = flat.sort(descending=True, stable=True)
            assert torch.equal(score.cpu(), v[:topk])
            got = idx.cpu().tolist()
            assert len(set(got)) == topk and all(float(flat[i]) == float(s) for i, s in zip(got, score.cpu()))
            for a, b in zip(got, got[1:]):                   # ties in ascending index order
                assert float(flat[a]) > float(flat[b]) or a < b
            cut = float(v[topk - 1])                         # every index below the last selected one with the cut value is in
            last = max(i for i in got if float(flat[i]) == cut)
            assert all((float(flat[i]) != cut) or (i in set(got)) for i in range(last))
        else:
            assert idx.cpu().tolist() == order, (case, Q, K, nrows, topk)
            assert torch.equal(score.cpu(), flat[order])
        assert sel_q.cpu().tolist() == [int(rows[i // K]) for i in idx.cpu().tolist()]


@pytest.mark.parametrize("sizes,scale", [([(4, 7), (8, 14), (15, 27)], 3.0), ([(23, 40), (46, 80), (92, 160)], 2.0),
                                         ([(23, 40), (46, 80), (92, 160)], 12.0), ([(5, 5), (9, 10), (17, 19)], 1.0)])
def test_msda_encoder_tiled_lds_kernel_is_identical_to_direct_gather(sizes, scale):
    """The LDS-staged tiled K1 (finest level) + direct gather (coarse levels) must give exactly the direct-gather result for
    any offsets -- small ones hit the staged windows, large ones (scale 12 px) take the global fallback."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(len(sizes) + int(scale))
    shapes = torch.tensor(sizes)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    B = 2
    value = torch.randn(B, S, 256, generator=g).cuda()
    oa = torch.randn(B, S, 288, generator=g)
    oa[..., :192] *= scale
    oa = oa.cuda()
    plain = ops.msda_encoder_fused(value, oa, shapes.cuda(), lsi.cuda())
    old = ops.MSDA_TILE_RADIUS
    try:
        for radius in (2, 4):
            ops.MSDA_TILE_RADIUS = radius                        # the tiled variant is off by default (slower, see ops.py)
            tiled = ops.msda_encoder_fused(value, oa, shapes.cuda(), lsi.cuda(), shapes_host=sizes)
            assert torch.equal(plain, tiled), radius
    finally:
        ops.MSDA_TILE_RADIUS = old


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("fp16", 3e-2)])
def test_clip_patch14_tower_and_crops_vs_oracle(precision, tol):
    """ViT-L/14-style geometry: patch 14 -> 588-column patch rows, padded to 592 (ops.patch_row_len) for 16-byte fp16 rows;
    crops (ClipAdapter) and the tower against the oracle at a small resolution (56 = 4 x 14)."""
    from openvis_amd import ops, weights
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from oracle import torch_ref as TR
    arch = dict(width=256, layers=2, heads=4, patch=14, resolution=56, embed_dim=64)
    sd = weights.random_init(weights.clip_visual_spec(**arch), seed=31)
    ad = ClipAdapter("tiny14", arch=arch, precision=precision).load_state_dict(sd, "clip_adapter.", "cuda")
    assert ops.patch_row_len(14) == 592
    g = torch.Generator().manual_seed(2)
    T, Q, H, W = 2, 5, 70, 90
    Hp, Wp = 96, 96
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8)
    masks = torch.randn(Q, T, Hp // 4, Wp // 4, generator=g) * 3
    names = [f"c{i}" for i in range(6)]
    text = torch.nn.functional.normalize(torch.randn(6, 64, generator=g), dim=-1)
    ad.set_text_features(names, text)
    logits, valid, crops = ad(frames.cuda(), names, masks.cuda(), (Hp, Wp))
    up = torch.nn.functional.interpolate(masks, size=(Hp, Wp), mode="bilinear", align_corners=False)
    part = up.sigmoid().transpose(0, 1).contiguous()                            # [T,Q,Hp,Wp]
    with torch.no_grad():
        regions, v2, _ = TR.clip_crops(frames, part, resolution=56)
        feat = TR.clip_encode_image(regions, sd, resolution=56, heads=4)
        ref = 100.0 * feat @ text.T
    assert (valid == v2.numpy()).all()
    assert (logits.cpu() - ref).abs().max().item() < tol * 100, (logits.cpu() - ref).abs().max().item()


def test_crop_kernel_merged_taps_and_out_of_frame_tiles():
    """clip_crop_tiled_kernel: (a) interior bins read a source pixel once for the frame AND the mask tap -- bit-identical to the two
    separate loops (lab switch ovis_crop_tile(16)); (b) tiles of a square roi that lie wholly below / right of the frame are written
    without staging -- the same values as the untiled kernel gives (0 * 0 -> the normalised zero pixel), on wide, tall and small boxes."""
    from openvis_amd import ops
    from openvis_amd import _lib
    T, Q, H, W, Hp, Wp = 2, 6, 360, 640, 384, 640
    g = torch.Generator().manual_seed(11)
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8).cuda()
    masks = (torch.randn(Q, T, Hp // 4, Wp // 4, generator=g) * 2).cuda()
    boxes = [[0, 0, 0, 0, Wp - 1, Hp - 1], [1, 1, 0, 0, W - 1, H - 1], [0, 2, 5, 7, 604, 100], [1, 3, 560, 3, 600, 350], [0, 4, 50, 60, 90, 95],
             [1, 5, 600, 300, 639, 359]]
    cr = torch.tensor(boxes, dtype=torch.int32).cuda()
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    try:
        merged = ops.clip_crop_patches(frames, masks, cr, Hp, Wp, 224, 16, mean, std, out_f16=False)
        _lib.call("ovis_crop_tile", 16)
        split = ops.clip_crop_patches(frames, masks, cr, Hp, Wp, 224, 16, mean, std, out_f16=False)
    finally:
        _lib.call("ovis_crop_tile", 0)
    assert torch.equal(merged, split)
    # wide box 0: the roi is 640 x 640 on a 384-row mask / 360-row frame -> bin rows >= 224 * 384 / 640 = 134.4 see no sample at all
    P = merged.view(len(boxes), 14, 14, 3, 16, 16)
    zero = ((torch.zeros(3) - torch.tensor(mean)) / torch.tensor(std)).cuda().view(1, 3, 1, 1)          # f32 arithmetic, as in the kernel
    assert torch.equal(P[0, 9:], zero.expand(5, 14, 3, 16, 16))
    assert not torch.equal(P[0, 7], P[0, 13])
    # tall box 3 (41 x 348 -> roi 348 wide from x = 560): bins from (640 - 560) / (348 / 224) = 51.5 on lie right of the frame
    assert torch.equal(P[3, :, 4:], zero.expand(14, 10, 3, 16, 16)) and not torch.equal(P[3, :, 3], P[3, :, 4])


@pytest.mark.parametrize("masked", [False, True])
def test_crop_leader_follower_passes_are_bit_identical_to_the_fused_pass(masked):
    """Round 6: crops of one frame that share a box compute the frame half (roi_align of the RGB planes, adapter.py:104-108) ONCE -- a
    leader pass that keeps the bins' frame averages and a follower pass that evaluates the mask half only (csrc/openvis_ops.hip,
    clip_crop_tiled_kernel MODE 1 / 2 + crop_dedupe_kernel).  Against the one-pass kernel (lab switch ovis_crop_tile(32)) on a crop list
    with duplicates inside a frame, the same box on ANOTHER frame (not a duplicate), unique boxes, a far-away (empty-mask) box twice and a
    leader that is not the first crop of its frame: every output bit equal, fp16 and f32, and patch_open of the masked variant too."""
    from openvis_amd import ops
    from openvis_amd import _lib
    T, Q, H, W, Hp, Wp = 3, 8, 360, 640, 384, 640
    g = torch.Generator().manual_seed(13)
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8).cuda()
    masks = (torch.randn(Q, T, Hp // 4, Wp // 4, generator=g) * 2).cuda()
    full, far = [0, 0, Wp - 1, Hp - 1], [4 * Wp + 4096, 4 * Hp + 4096, 4 * Wp + 4096, 4 * Hp + 4096]
    boxes = [[0, 0] + full, [0, 1, 5, 7, 604, 100], [0, 2] + full, [0, 3, 5, 7, 604, 100], [0, 4] + full, [0, 5] + far, [0, 6] + far, [0, 7, 50, 60, 90, 95],
             [1, 0, 50, 60, 90, 95], [1, 1] + full, [1, 2, 0, 0, W - 1, H - 1], [1, 3] + full, [1, 5, 0, 0, W - 1, H - 1],
             [2, 0, 560, 3, 600, 350], [2, 4, 560, 3, 600, 350], [2, 7, 560, 3, 600, 351]]
    cr = torch.tensor(boxes, dtype=torch.int32).cuda()
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    fn = ops.clip_crop_patches_masked if masked else ops.clip_crop_patches
    for f16 in (False, True):
        try:
            two = fn(frames, masks, cr, Hp, Wp, 224, 16, mean, std, out_f16=f16)
            _lib.call("ovis_crop_tile", 32)
            one = fn(frames, masks, cr, Hp, Wp, 224, 16, mean, std, out_f16=f16)
        finally:
            _lib.call("ovis_crop_tile", 0)
        if masked:
            assert torch.equal(two[1], one[1]) and 0 < int(one[1].sum()) < one[1].numel()
            two, one = two[0], one[0]
        assert torch.equal(two, one)
        P = one.view(len(boxes), 14 * 14, -1)
        assert not torch.equal(P[0], P[2]) and not torch.equal(P[0], P[9])          # same box, other query / other frame: different crops


def test_preprocess_whole_video_beyond_65535_rows():
    """A1 on a whole video (openvis.py:57-62 runs before the windowing, :109): T * Hp = 720 * 96 = 69 120 rows > 65 535, the y limit of a HIP
    grid -- the rows sit on gridDim.x.  Bit-exact against the oracle's arithmetic ((x - mean) / std in f32, zero pad)."""
    from openvis_amd import ops
    from oracle import torch_ref as TR
    g = torch.Generator().manual_seed(5)
    T, H, W = 720, 90, 310                                     # Hp = 96, Wp = 320
    frames = torch.randint(0, 256, (T, 3, H, W), generator=g, dtype=torch.uint8)
    out = ops.preprocess_u8(frames.cuda(), 96, 320, TR.PIXEL_MEAN, TR.PIXEL_STD)
    ref, _ = TR.preprocess([f for f in frames])
    assert tuple(out.shape) == (T, 96, 320, 4) and T * 96 > 65535
    assert torch.equal(out[..., :3].permute(0, 3, 1, 2).cpu(), ref) and out[..., 3].abs().max().item() == 0


@pytest.mark.parametrize("rows", [100, 501, 3])
def test_ln_mlp3_matches_layernorm_and_three_linears(rows):
    """ovis_ln_mlp3_f32 (decoder_norm + mask-embedding MLP of forward_prediction_heads, video_mask2former_transformer_decoder.py:454-458, one
    launch) against torch in f64: both outputs, row counts that are no multiple of the 4 rows a workgroup owns."""
    import torch.nn.functional as F
    from openvis_amd import ops
    g = torch.Generator().manual_seed(rows)
    C = 256
    x = (torch.randn(rows, C, generator=g) * 3 + 0.5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ws = [torch.randn(C, C, generator=g) / 16 for _ in range(3)]
    bs = [torch.randn(C, generator=g) for _ in range(3)]
    dec_ref = F.layer_norm(x.double(), (C,), gamma.double(), beta.double())
    h = dec_ref
    for j in range(3):
        h = h @ ws[j].double().t() + bs[j].double()
        if j < 2:
            h = h.relu()
    dec, out = ops.ln_mlp3(x.cuda(), gamma.cuda(), beta.cuda(), [w.t().contiguous().cuda() for w in ws], [b.cuda() for b in bs])
    torch.cuda.synchronize()
    assert (dec.cpu().double() - dec_ref).abs().max().item() < 1e-5
    assert (out.cpu().double() - h).abs().max().item() < 1e-4 * h.abs().max().item()
    _, out2 = ops.ln_mlp3(x.cuda(), gamma.cuda(), beta.cuda(), [w.t().contiguous().cuda() for w in ws], [b.cuda() for b in bs], want_dec=False)
    assert torch.equal(out2, out)
