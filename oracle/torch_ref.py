"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain torch, fp32/fp64) of the reference's
per-frame dense inference path for the OpenVIS meta-architecture.

Never imported by the product package.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg use it, as the checker / the timed CPU baseline ("port").

Each function cites the reference file:line it follows (paths relative to /root/reference/).
State-dict key names are the reference's.  Pinning status:
  * pixel decoder, decoders, position encodings, CLIP visual tower: PINNED — checked against
    the reference's own modules imported in the build container (oracle/make_golden.py ->
    tests/golden/*.npz; tests/test_oracle_*.py).
  * detectron2 ResNet-50 / ImageList / BitMasks boxes, torchvision roi_align: the sources are NOT
    under /root/reference (un-vendored third-party deps: detectron2 v0.6 per INSTALL.md:9-13,
    torchvision 0.11) -> restated from their published semantics; "parity unpinned" for those rows.
  * dtype: the reference's GPU path casts crops to fp16 (clip_adapter/adapter.py:108-111: `.half()` on frames,
    boxes and the soft mask) and runs under autocast; this oracle follows the fp32 semantics of the same
    statements with the `.cuda()` / `.half()` casts dropped ("the fp32 oracle must state which
    it follows" -- crops, roi_align and the blend are f32 here).  ONE consequence of the fp16 cast is
    discrete and is restated: the mask-prompt tower opens a patch iff its pooled mask region is > 0
    (model.py:332-333), and in fp16 a soft-mask value below 2^-25 (logit < -17.3) is exactly 0 -- that is what
    closes background patches on a real checkpoint.  clip_visual() therefore rounds the mask regions to fp16
    before pooling them (mask_adapted_adapter.py:113).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# A1  pre-processing: openvis/openvis.py:57-62 (video_maskformer.py:178-183); Base.yaml:6-7
# ----------------------------------------------------------------------------------------------
PIXEL_MEAN = (123.675, 116.280, 103.530)
PIXEL_STD = (58.395, 57.120, 57.375)


def preprocess(frames, size_divisibility=32, mean=PIXEL_MEAN, std=PIXEL_STD):
    """frames: list/tensor of uint8 [3,H,W] -> (fp32 [T,3,Hp,Wp] zero-padded right/bottom, (H,W)).
    detectron2 ImageList.from_tensors(tensors, size_divisibility): pad to ceil(max/div)*div, value 0."""
    mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    imgs = [(x - mean) / std for x in frames]
    H = max(i.shape[-2] for i in imgs)
    W = max(i.shape[-1] for i in imgs)
    Hp = (H + size_divisibility - 1) // size_divisibility * size_divisibility
    Wp = (W + size_divisibility - 1) // size_divisibility * size_divisibility
    out = imgs[0].new_zeros(len(imgs), 3, Hp, Wp)
    for i, im in enumerate(imgs):
        out[i, :, : im.shape[-2], : im.shape[-1]] = im
    return out, (imgs[0].shape[-2], imgs[0].shape[-1])


# ----------------------------------------------------------------------------------------------
# A2  ResNet-50 backbone — detectron2 build_resnet_backbone (configs/openvoc_ytvis_coco/Base.yaml:2-16:
#     DEPTH 50, STRIDE_IN_1X1 False, FrozenBN default, out res2..res5).  UN-VENDORED: restated from
#     detectron2 v0.6 modeling/backbone/resnet.py (BasicStem, BottleneckBlock) + layers/batch_norm.py
#     (FrozenBatchNorm2d, eps 1e-5).  parity unpinned.
# ----------------------------------------------------------------------------------------------
def _frozen_bn(x, W, prefix, eps=1e-5):
    scale = W[prefix + ".weight"] * (W[prefix + ".running_var"] + eps).rsqrt()
    bias = W[prefix + ".bias"] - W[prefix + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def _conv_bn(x, W, prefix, stride=1, padding=0):
    x = F.conv2d(x, W[prefix + ".weight"], None, stride=stride, padding=padding)
    return _frozen_bn(x, W, prefix + ".norm")


RESNET50_STAGES = (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2), ("res5", 3, 2))


# ---- A2 (alternative backbone): Swin Transformer, backbone/swin.py -------------------------------------------------
def _swin_rel_index(ws):
    """relative_position_index of WindowAttention (swin.py:111-122)."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)      # [2, N]
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[..., 0] * (2 * ws - 1) + rel[..., 1]


def _swin_shift_mask(Hp, Wp, ws, shift):
    """0 / -100 attention mask of the shifted windows (swin.py:381-404)."""
    img = torch.zeros(Hp, Wp)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = mw[:, None, :] - mw[:, :, None]
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def swin(x, W, embed_dim, depths, num_heads, ws, prefix="backbone.", patch_norm=True):
    """x fp32 [B,3,H,W] -> {res2..res5: [B,C_i,H_i,W_i]} (swin.py:702-722, 224-284, 303-317, 480-497)."""
    x = F.conv2d(x, W[prefix + "patch_embed.proj.weight"], W[prefix + "patch_embed.proj.bias"], stride=4)
    B, C, H, Wd = x.shape
    x = x.flatten(2).transpose(1, 2)                                                             # [B, H*W, C]
    if patch_norm:
        x = F.layer_norm(x, (C,), W[prefix + "patch_embed.norm.weight"], W[prefix + "patch_embed.norm.bias"])
    if prefix + "absolute_pos_embed" in W:                                                       # ape=True (swin.py:567-578, 706-713)
        ape = F.interpolate(W[prefix + "absolute_pos_embed"], size=(H, Wd), mode="bicubic")
        x = x + ape.flatten(2).transpose(1, 2)
    outs = {}
    for i, depth in enumerate(depths):
        C = embed_dim * 2 ** i
        heads = num_heads[i]
        Hp, Wp = -(-H // ws) * ws, -(-Wd // ws) * ws
        shift_mask = _swin_shift_mask(Hp, Wp, ws, ws // 2)
        ridx = _swin_rel_index(ws).view(-1)
        for j in range(depth):
            p = f"{prefix}layers.{i}.blocks.{j}."
            shift = 0 if j % 2 == 0 else ws // 2
            h = F.layer_norm(x, (C,), W[p + "norm1.weight"], W[p + "norm1.bias"]).view(B, H, Wd, C)
            h = F.pad(h, (0, 0, 0, Wp - Wd, 0, Hp - H))
            if shift:
                h = torch.roll(h, (-shift, -shift), (1, 2))
            win = h.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
            qkv = F.linear(win, W[p + "attn.qkv.weight"], W.get(p + "attn.qkv.bias"))
            qkv = qkv.view(win.shape[0], ws * ws, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
            attn = (qkv[0] * (C // heads) ** -0.5) @ qkv[1].transpose(-2, -1)
            bias = W[p + "attn.relative_position_bias_table"][ridx].view(ws * ws, ws * ws, heads).permute(2, 0, 1)
            attn = attn + bias[None]
            if shift:
                nW = shift_mask.shape[0]
                attn = (attn.view(B, nW, heads, ws * ws, ws * ws) + shift_mask[None, :, None]).view(-1, heads, ws * ws, ws * ws)
            a = (attn.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(-1, ws * ws, C)
            a = F.linear(a, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])
            a = a.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
            if shift:
                a = torch.roll(a, (shift, shift), (1, 2))
            x = x + a[:, :H, :Wd].reshape(B, H * Wd, C)
            h = F.layer_norm(x, (C,), W[p + "norm2.weight"], W[p + "norm2.bias"])
            h = F.linear(F.gelu(F.linear(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])), W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
            x = x + h
        o = F.layer_norm(x, (C,), W[f"{prefix}norm{i}.weight"], W[f"{prefix}norm{i}.bias"])
        outs[f"res{i + 2}"] = o.view(B, H, Wd, C).permute(0, 3, 1, 2).contiguous()
        if i < len(depths) - 1:
            h = F.pad(x.view(B, H, Wd, C), (0, 0, 0, Wd % 2, 0, H % 2))
            h = torch.cat([h[:, 0::2, 0::2], h[:, 1::2, 0::2], h[:, 0::2, 1::2], h[:, 1::2, 1::2]], -1)
            H, Wd = (H + 1) // 2, (Wd + 1) // 2
            h = F.layer_norm(h.view(B, H * Wd, 4 * C), (4 * C,), W[f"{prefix}layers.{i}.downsample.norm.weight"],
                             W[f"{prefix}layers.{i}.downsample.norm.bias"])
            x = F.linear(h, W[f"{prefix}layers.{i}.downsample.reduction.weight"])
    return outs


def _h(x):
    """one fp16 rounding (what a half tensor holds), carried in f32 so that the CPU convolutions accumulate in f32 like the GPU's fp16 kernels"""
    return x.half().float()


def _conv_bn_autocast(x, W, prefix, stride=1, padding=0, eps=1e-5):
    """conv + FrozenBatchNorm2d as the reference's GPU eval runs them under `torch.cuda.amp.autocast` (train_net.py:241): conv2d is on
    autocast's fp16 list -- operands cast to fp16, f32 accumulation, fp16 result -- and detectron2 v0.6 FrozenBatchNorm2d.forward (eval
    branch) then computes `x * scale.to(x.dtype) + bias.to(x.dtype)` on that fp16 tensor: scale and bias rounded to fp16, the product
    rounded to fp16, the sum rounded to fp16.  x arrives holding fp16 values (or the f32 images: autocast casts them)."""
    y = _h(F.conv2d(_h(x), _h(W[prefix + ".weight"]), None, stride=stride, padding=padding))
    scale = W[prefix + ".norm.weight"] * (W[prefix + ".norm.running_var"] + eps).rsqrt()
    bias = W[prefix + ".norm.bias"] - W[prefix + ".norm.running_mean"] * scale
    return _h(_h(y * _h(scale).reshape(1, -1, 1, 1)) + _h(bias).reshape(1, -1, 1, 1))


def resnet50_autocast(x, W, prefix="backbone."):
    """resnet50() in the arithmetic of the reference's own GPU run: EVERY tensor between two ops is fp16 (conv outputs, BN outputs, ReLU,
    max pool, the residual sums `out += shortcut`), res2..res5 leave as fp16 values and the pixel decoder (autocast disabled,
    msdeformattn.py:329) converts them with .float().  This is the arithmetic the product's fp16 backbone policy stands in for; the product
    itself rounds LESS (BN folded into the fp16 weights, f32 accumulators carry bias / residual / ReLU, the block outputs stay f32) --
    oracle/make_golden_workload.py c2a measures all three against each other."""
    x = F.relu(_conv_bn_autocast(x, W, prefix + "stem.conv1", stride=2, padding=3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = {}
    for name, nblocks, first_stride in RESNET50_STAGES:
        for i in range(nblocks):
            p = f"{prefix}{name}.{i}"
            stride = first_stride if i == 0 else 1
            sc = _conv_bn_autocast(x, W, p + ".shortcut", stride=stride) if (p + ".shortcut.weight") in W else x
            out = F.relu(_conv_bn_autocast(x, W, p + ".conv1", stride=1))
            out = F.relu(_conv_bn_autocast(out, W, p + ".conv2", stride=stride, padding=1))
            out = _conv_bn_autocast(out, W, p + ".conv3")
            x = F.relu(_h(out + sc))                                          # `out += shortcut` on fp16 tensors
        feats[name] = x
    return feats


def resnet50(x, W, prefix="backbone."):
    x = F.relu(_conv_bn(x, W, prefix + "stem.conv1", stride=2, padding=3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = {}
    for name, nblocks, first_stride in RESNET50_STAGES:
        for i in range(nblocks):
            p = f"{prefix}{name}.{i}"
            stride = first_stride if i == 0 else 1
            if (p + ".shortcut.weight") in W:
                sc = _conv_bn(x, W, p + ".shortcut", stride=stride)
            else:
                sc = x
            out = F.relu(_conv_bn(x, W, p + ".conv1", stride=1))            # STRIDE_IN_1X1 False
            out = F.relu(_conv_bn(out, W, p + ".conv2", stride=stride, padding=1))
            out = _conv_bn(out, W, p + ".conv3")
            x = F.relu(out + sc)
        feats[name] = x
    return feats


# ----------------------------------------------------------------------------------------------
# A3  position encodings
# ----------------------------------------------------------------------------------------------
def pe_sine_2d(B, H, W, num_pos_feats=128, temperature=10000, scale=2 * math.pi, dtype=torch.float32):
    """openvis/modeling/pixel_decoder/position_encoding.py:29-53 (normalize=True, mask all-False)."""
    y_embed = torch.arange(1, H + 1, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
    x_embed = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = dim_t.div(2, rounding_mode="floor")
    dim_t = temperature ** (2 * dim_t / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).to(dtype)


def pe_sine_3d(B, T, H, W, num_pos_feats=128, temperature=10000, scale=2 * math.pi):
    """openvis/modeling/transformer_decoder/position_encoding.py:135-165 -> [B,T,2*npf,H,W]."""
    z_embed = torch.arange(1, T + 1, dtype=torch.float32).view(1, T, 1, 1).expand(B, T, H, W)
    y_embed = torch.arange(1, H + 1, dtype=torch.float32).view(1, 1, H, 1).expand(B, T, H, W)
    x_embed = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, 1, W).expand(B, T, H, W)
    eps = 1e-6
    z_embed = z_embed / (z_embed[:, -1:, :, :] + eps) * scale
    y_embed = y_embed / (y_embed[:, :, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32).div(2, rounding_mode="floor")
    dim_t = temperature ** (2 * dim_t / num_pos_feats)
    dim_t_z = torch.arange(num_pos_feats * 2, dtype=torch.float32).div(2, rounding_mode="floor")
    dim_t_z = temperature ** (2 * dim_t_z / (num_pos_feats * 2))
    pos_x = x_embed[..., None] / dim_t
    pos_y = y_embed[..., None] / dim_t
    pos_z = z_embed[..., None] / dim_t_z
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=5).flatten(4)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=5).flatten(4)
    pos_z = torch.stack((pos_z[..., 0::2].sin(), pos_z[..., 1::2].cos()), dim=5).flatten(4)
    return (torch.cat((pos_y, pos_x), dim=4) + pos_z).permute(0, 1, 4, 2, 3)


# ----------------------------------------------------------------------------------------------
# A5  K1 in torch (same tap rules / accumulation order as oracle/msda_ref.c), any float dtype
# ----------------------------------------------------------------------------------------------
def msda_torch(value, shapes, lsi, loc, attw):
    """ms_deform_im2col_cuda.cuh:242-304 restated with gathers; l-then-p accumulation order."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = value.new_zeros(B, Lq, M, D)
    bidx = torch.arange(B).view(B, 1, 1).expand(B, Lq, M)
    midx = torch.arange(M).view(1, 1, M).expand(B, Lq, M)
    for l in range(L):
        H, W = int(shapes[l][0]), int(shapes[l][1])
        start = int(lsi[l])
        v = value[:, start:start + H * W]
        for p in range(P):
            w_im = loc[:, :, :, l, p, 0] * W - 0.5
            h_im = loc[:, :, :, l, p, 1] * H - 0.5
            inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            h_low = torch.floor(h_im)
            w_low = torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            hh, hw = 1 - lh, 1 - lw
            h_low, w_low = h_low.long(), w_low.long()
            h_high, w_high = h_low + 1, w_low + 1

            def tap(hi, wi):
                ok = (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)
                idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1))
                g = v[bidx, idx, midx]  # [B,Lq,M,D]
                return g * ok[..., None].to(g.dtype)

            v1, v2, v3, v4 = tap(h_low, w_low), tap(h_low, w_high), tap(h_high, w_low), tap(h_high, w_high)
            w1, w2, w3, w4 = (hh * hw)[..., None], (hh * lw)[..., None], (lh * hw)[..., None], (lh * lw)[..., None]
            val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4)
            out = out + (val * attw[:, :, :, l, p, None]) * inside[..., None].to(val.dtype)
    return out.reshape(B, Lq, M * D)


# ----------------------------------------------------------------------------------------------
# A3-A6  MSDeformAttnPixelDecoder.forward_features — openvis/modeling/pixel_decoder/msdeformattn.py:329-380
# ----------------------------------------------------------------------------------------------
def _gn(x, W, prefix, groups=32):
    return F.group_norm(x, groups, W[prefix + ".weight"], W[prefix + ".bias"], eps=1e-5)


def msdeform_attn(W, p, query, ref_points, src, shapes, lsi, n_heads=8, n_levels=3, n_points=4):
    """ops/modules/ms_deform_attn.py:82-125."""
    N, Lq, C = query.shape
    value = F.linear(src, W[p + "value_proj.weight"], W[p + "value_proj.bias"]).view(N, -1, n_heads, C // n_heads)
    off = F.linear(query, W[p + "sampling_offsets.weight"], W[p + "sampling_offsets.bias"]).view(
        N, Lq, n_heads, n_levels, n_points, 2)
    aw = F.linear(query, W[p + "attention_weights.weight"], W[p + "attention_weights.bias"]).view(
        N, Lq, n_heads, n_levels * n_points)
    aw = F.softmax(aw, -1).view(N, Lq, n_heads, n_levels, n_points)
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_torch(value, shapes, lsi, loc, aw)
    return F.linear(out, W[p + "output_proj.weight"], W[p + "output_proj.bias"])


def encoder_reference_points(shapes_list):
    """msdeformattn.py:155-168 with valid_ratios == 1 -> [1,S,L,2]."""
    refs = []
    for (H_, W_) in shapes_list:
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32),
                                      torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32), indexing="ij")
        ref_y = ref_y.reshape(-1)[None] / H_
        ref_x = ref_x.reshape(-1)[None] / W_
        refs.append(torch.stack((ref_x, ref_y), -1))
    ref = torch.cat(refs, 1)
    return ref[:, :, None].expand(-1, -1, len(shapes_list), -1)


def pixel_decoder(feats, W, prefix="sem_seg_head.pixel_decoder.", n_layers=6, return_intermediate=False,
                  extra_features=None):
    """feats: dict res2..res5 (fp32 NCHW) -> (mask_features, out[0], multi_scale_features[3])."""
    p = prefix
    srcs, poss = [], []
    for idx, f in enumerate(["res5", "res4", "res3"]):                       # msdeformattn.py:332-337
        x = feats[f].float()
        y = F.conv2d(x, W[f"{p}input_proj.{idx}.0.weight"], W[f"{p}input_proj.{idx}.0.bias"])
        srcs.append(_gn(y, W, f"{p}input_proj.{idx}.1"))
        poss.append(pe_sine_2d(x.shape[0], x.shape[2], x.shape[3]))
        if extra_features is not None:                                          # msdeformattn.py:338-344 (SAN)
            ex = extra_features[idx]
            if ex.shape[-2:] != x.shape[-2:]:
                ex = F.interpolate(ex, size=x.shape[-2:], mode="bilinear", align_corners=False)
            srcs[-1] = srcs[-1] + ex
    # MSDeformAttnTransformerEncoderOnly.forward: msdeformattn.py:76-104
    shapes_list = [(s.shape[2], s.shape[3]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([q.flatten(2).transpose(1, 2) + W[p + "transformer.level_embed"][l].view(1, 1, -1)
                     for l, q in enumerate(poss)], 1)
    shapes = torch.as_tensor(shapes_list, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    ref = encoder_reference_points(shapes_list).expand(src.shape[0], -1, -1, -1)
    inter = {"src0": src, "pos": pos}
    out = src
    for i in range(n_layers):                                               # msdeformattn.py:137-146
        lp = f"{p}transformer.encoder.layers.{i}."
        src2 = msdeform_attn(W, lp + "self_attn.", out + pos, ref, out, shapes, lsi)
        out = F.layer_norm(out + src2, (256,), W[lp + "norm1.weight"], W[lp + "norm1.bias"])
        src2 = F.linear(F.relu(F.linear(out, W[lp + "linear1.weight"], W[lp + "linear1.bias"])),
                        W[lp + "linear2.weight"], W[lp + "linear2.bias"])
        out = F.layer_norm(out + src2, (256,), W[lp + "norm2.weight"], W[lp + "norm2.bias"])
    inter["memory"] = out
    bs = out.shape[0]
    ys = torch.split(out, [h * w for h, w in shapes_list], dim=1)            # msdeformattn.py:349-362
    outs = [z.transpose(1, 2).reshape(bs, -1, h, w) for z, (h, w) in zip(ys, shapes_list)]
    # FPN on res2: msdeformattn.py:365-373
    x = feats["res2"].float()
    cur = _gn(F.conv2d(x, W[p + "adapter_1.weight"]), W, p + "adapter_1.norm")
    y = cur + F.interpolate(outs[-1], size=cur.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(_gn(F.conv2d(y, W[p + "layer_1.weight"], padding=1), W, p + "layer_1.norm"))
    outs.append(y)
    mask_features = F.conv2d(outs[-1], W[p + "mask_features.weight"], W[p + "mask_features.bias"])
    if return_intermediate:
        return mask_features, outs[0], outs[:3], inter
    return mask_features, outs[0], outs[:3]


# ----------------------------------------------------------------------------------------------
# A7/A8  VideoMultiScaleMaskedTransformerDecoder — transformer_decoder/video_mask2former_transformer_decoder.py:380-471
# ----------------------------------------------------------------------------------------------
def _mha(W, p, query, key, value, attn_mask=None, nheads=8):
    """nn.MultiheadAttention (seq-first) via torch's own functional definition."""
    out, _ = F.multi_head_attention_forward(
        query, key, value, query.shape[-1], nheads, W[p + "in_proj_weight"], W[p + "in_proj_bias"], None, None,
        False, 0.0, W[p + "out_proj.weight"], W[p + "out_proj.bias"], training=False, key_padding_mask=None,
        need_weights=False, attn_mask=attn_mask)
    return out


def _ln(x, W, p):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"])


def _mlp3(x, W, p):
    x = F.relu(F.linear(x, W[p + "layers.0.weight"], W[p + "layers.0.bias"]))
    x = F.relu(F.linear(x, W[p + "layers.1.weight"], W[p + "layers.1.bias"]))
    return F.linear(x, W[p + "layers.2.weight"], W[p + "layers.2.bias"])


def prediction_heads(W, p, output, mask_features, target_size, nheads=8):
    """video decoder:454-471. output [Q,bs,C]; mask_features [bs,T,C,H,W]."""
    dec = _ln(output, W, p + "decoder_norm").transpose(0, 1)
    outputs_class = F.linear(dec, W[p + "class_embed.weight"], W[p + "class_embed.bias"])
    mask_embed = _mlp3(dec, W, p + "mask_embed.")
    outputs_mask = torch.einsum("bqc,btchw->bqthw", mask_embed, mask_features)
    b, q, t, _, _ = outputs_mask.shape
    am = F.interpolate(outputs_mask.flatten(0, 1), size=target_size, mode="bilinear", align_corners=False).view(
        b, q, t, target_size[0], target_size[1])
    am = (am.sigmoid().flatten(2).unsqueeze(1).repeat(1, nheads, 1, 1).flatten(0, 1) < 0.5).bool()
    return outputs_class, outputs_mask, am


def video_decoder(ms_feats, mask_features, W, prefix="sem_seg_head.predictor.", n_layers=9, nheads=8,
                  return_intermediate=False):
    """eval mode: bs = 1, t = T (video decoder:381-383). Returns pred_logits [1,Q,C+1], pred_masks [1,Q,T,H,W]."""
    p = prefix
    bt, c_m, h_m, w_m = mask_features.shape
    bs, t = 1, bt
    mask_features = mask_features.view(bs, t, c_m, h_m, w_m)
    src, pos, size_list = [], [], []
    for i in range(3):
        h, w = ms_feats[i].shape[-2:]
        size_list.append((h, w))
        pe = pe_sine_3d(bs, t, h, w).flatten(3)                               # [bs,t,c,hw]
        s = ms_feats[i].flatten(2) + W[p + "level_embed.weight"][i][None, :, None]
        c = s.shape[1]
        pos.append(pe.view(bs, t, c, h * w).permute(1, 3, 0, 2).flatten(0, 1))
        src.append(s.view(bs, t, c, h * w).permute(1, 3, 0, 2).flatten(0, 1))
    query_embed = W[p + "query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    output = W[p + "query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)
    inter = []
    cls, msk, attn_mask = prediction_heads(W, p, output, mask_features, size_list[0], nheads)
    inter.append(msk)
    for i in range(n_layers):
        li = i % 3
        attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False   # video decoder:419
        cp = f"{p}transformer_cross_attention_layers.{i}."
        tgt2 = _mha(W, cp + "multihead_attn.", output + query_embed, src[li] + pos[li], src[li], attn_mask, nheads)
        output = _ln(output + tgt2, W, cp + "norm")
        sp = f"{p}transformer_self_attention_layers.{i}."
        qk = output + query_embed
        tgt2 = _mha(W, sp + "self_attn.", qk, qk, output, None, nheads)
        output = _ln(output + tgt2, W, sp + "norm")
        fp = f"{p}transformer_ffn_layers.{i}."
        tgt2 = F.linear(F.relu(F.linear(output, W[fp + "linear1.weight"], W[fp + "linear1.bias"])),
                        W[fp + "linear2.weight"], W[fp + "linear2.bias"])
        output = _ln(output + tgt2, W, fp + "norm")
        cls, msk, attn_mask = prediction_heads(W, p, output, mask_features, size_list[(i + 1) % 3], nheads)
        inter.append(msk)
    if return_intermediate:
        return cls, msk, inter, output
    return cls, msk


# ----------------------------------------------------------------------------------------------
# A10  ClipAdapter crops — openvis/modeling/clip_adapter/adapter.py:73-116
# ----------------------------------------------------------------------------------------------
def bitmask_boxes(bin_masks):
    """detectron2 BitMasks.get_bounding_boxes (v0.6 structures/masks.py): [x0, y0, x1+1, y1+1]; zeros if empty.
    UN-VENDORED -> restated; parity unpinned."""
    n = bin_masks.shape[0]
    boxes = torch.zeros(n, 4, dtype=torch.float32)
    x_any = torch.any(bin_masks, dim=1)
    y_any = torch.any(bin_masks, dim=2)
    for idx in range(n):
        x = torch.where(x_any[idx, :])[0]
        y = torch.where(y_any[idx, :])[0]
        if len(x) > 0 and len(y) > 0:
            boxes[idx, :] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
    return boxes


def roi_align(inp, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """torchvision.ops.roi_align (0.11, csrc/ops/cpu/roi_align_kernel.cpp) restated; rois [K,5] = (batch, x1,y1,x2,y2).
    UN-VENDORED -> parity unpinned. Vectorised over the adaptive sampling grid per roi."""
    K = rois.shape[0]
    C, H, W = inp.shape[1:]
    ph, pw = output_size
    out = inp.new_zeros(K, C, ph, pw)
    offset = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(rois[k, 0])
        x1, y1, x2, y2 = [float(v) * spatial_scale - offset for v in rois[k, 1:]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        count = max(gh * gw, 1)
        dt = torch.float64 if inp.dtype == torch.float64 else torch.float32
        iy = (torch.arange(gh, dtype=dt) + 0.5) * bh / gh
        ix = (torch.arange(gw, dtype=dt) + 0.5) * bw / gw
        ys = (y1 + torch.arange(ph, dtype=dt)[:, None] * bh + iy[None, :]).reshape(-1)   # [ph*gh]
        xs = (x1 + torch.arange(pw, dtype=dt)[:, None] * bw + ix[None, :]).reshape(-1)   # [pw*gw]

        def prep(v, size):
            bad = (v < -1.0) | (v > size)
            v = v.clamp(min=0)
            lo = v.floor().long()
            hi_edge = lo >= size - 1
            lo = torch.where(hi_edge, torch.full_like(lo, size - 1), lo)
            hi = torch.where(hi_edge, lo, lo + 1)
            v = torch.where(hi_edge, lo.to(v.dtype), v)
            l = v - lo.to(v.dtype)
            return lo, hi, l, 1 - l, bad

        ylo, yhi, ly, hy, ybad = prep(ys, H)
        xlo, xhi, lx, hx, xbad = prep(xs, W)
        img = inp[b]
        v1 = img[:, ylo][:, :, xlo]
        v2 = img[:, ylo][:, :, xhi]
        v3 = img[:, yhi][:, :, xlo]
        v4 = img[:, yhi][:, :, xhi]
        w1 = (hy[:, None] * hx[None, :]).to(inp.dtype)
        w2 = (hy[:, None] * lx[None, :]).to(inp.dtype)
        w3 = (ly[:, None] * hx[None, :]).to(inp.dtype)
        w4 = (ly[:, None] * lx[None, :]).to(inp.dtype)
        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
        val = val * (~(ybad[:, None] | xbad[None, :])).to(inp.dtype)
        out[k] = val.view(C, ph, gh, pw, gw).sum(dim=(2, 4)) / count
    return out


def clip_crops(frames, masks, resolution=224, return_mask_regions=False):
    """adapter.py:73-116 (== mask_adapted_adapter.py:79-123, which also returns mask_regions). frames [T,3,H,W] (raw
    0..255, un-padded), masks [T,N,Hp,Wp] (sigmoid probabilities).
    Returns (regions [M,3,res,res] or None, valid [T,N] bool, boxes [M,4]) (+ mask_regions [M,1,res,res])."""
    frames = frames.float()
    bin_masks = masks > 0.5
    valid = bin_masks.sum(dim=(-1, -2)) > 0
    if torch.sum(valid) == 0:
        return (None, valid, None, None) if return_mask_regions else (None, valid, None)
    valid_bin_masks = bin_masks[valid]
    valid_masks = masks[valid]
    sboxes = bitmask_boxes(valid_bin_masks).clone()
    sboxes[:, 2] = sboxes[:, 2] - sboxes[:, 0]
    sboxes[:, 3] = sboxes[:, 3] - sboxes[:, 1]
    sboxes[:, 3] = sboxes[:, 2] = torch.max(sboxes[:, 2], sboxes[:, 3])
    sboxes[:, 2] = sboxes[:, 0] + sboxes[:, 2]
    sboxes[:, 3] = sboxes[:, 1] + sboxes[:, 3]
    ids = torch.nonzero(valid)
    ind = torch.cat([ids[:, 0:1].float(), sboxes], dim=-1)
    regions = roi_align(frames, ind, (resolution, resolution))
    ind = torch.cat([torch.arange(len(sboxes))[:, None].float(), sboxes], dim=-1)
    mask_regions = roi_align(valid_masks[:, None], ind, (resolution, resolution))
    regions = mask_regions * regions + (1 - mask_regions) * 0.
    if return_mask_regions:
        return regions, valid, sboxes, mask_regions
    return regions, valid, sboxes


# ----------------------------------------------------------------------------------------------
# A10  CLIP visual tower (ViT) — third_parties/mask_adapted_clip/mask_adapted_clip/model.py:327-362 with m=None
#      (identical to openai/CLIP VisionTransformer.forward), blocks model.py:238-268, LayerNorm 223-229, QuickGELU 232-234
# ----------------------------------------------------------------------------------------------
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_visual(x, W, prefix="clip_adapter.clip_model.visual.", heads=12, m=None, mask_prompt_depth=0):
    """m: None (model.py:327-362 with m=None) or mask regions [M,1,res,res] — the mask-prompt path (model.py:331-354):
    patch tokens whose pooled mask is 0 are replaced by mask_embedding[0] before the class token / positional embedding,
    and again by mask_embedding[d] after block d for d < mask_prompt_depth."""
    p = prefix
    patch = W[p + "conv1.weight"].shape[-1]
    x = F.conv2d(x, W[p + "conv1.weight"], None, stride=patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    if m is not None:
        m = m.half().float()     # the reference's mask regions are fp16 (mask_adapted_adapter.py:113): < 2^-25 is 0
        m = F.avg_pool2d(m.reshape(m.shape[0], 1, m.shape[-2], m.shape[-1]), patch, stride=patch)
        m = torch.ceil(m.reshape(m.shape[0], -1).unsqueeze(-1))                # [M, G*G, 1]
        mask_embedding = W[p + "mask_embedding"]
        if mask_embedding.shape[1] == 1:
            mask_embedding = mask_embedding.repeat(1, x.shape[1], 1)
        x = x * m + mask_embedding[0].unsqueeze(0) * (1 - m)
    cls = W[p + "class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + W[p + "positional_embedding"]
    x = _ln(x, W, p + "ln_pre")
    x = x.permute(1, 0, 2)
    n_layers = 1 + max(int(k[len(p + "transformer.resblocks."):].split(".")[0]) for k in W
                       if k.startswith(p + "transformer.resblocks."))
    for i in range(n_layers):
        bp = f"{p}transformer.resblocks.{i}."
        h = _ln(x, W, bp + "ln_1")
        x = x + _mha(W, bp + "attn.", h, h, h, None, heads)
        h = _ln(x, W, bp + "ln_2")
        h = F.linear(h, W[bp + "mlp.c_fc.weight"], W[bp + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, W[bp + "mlp.c_proj.weight"], W[bp + "mlp.c_proj.bias"])
        if m is not None and i + 1 < mask_prompt_depth:                        # model.py:349-352
            mp = m.permute(1, 0, 2)
            masked_x = x[1:] * mp + mask_embedding[i + 1].unsqueeze(0).permute(1, 0, 2) * (1 - mp)
            x = torch.cat([x[:1], masked_x], dim=0)
    x = x.permute(1, 0, 2)
    x = _ln(x[:, 0, :], W, p + "ln_post")
    return x @ W[p + "proj"]


def clip_encode_text(tokens, W, prefix="clip_adapter.clip_model.", heads=8):
    """CLIP.encode_text (mask_adapted_clip/model.py:478-491): token + positional embedding, causal transformer,
    ln_final, feature of the eot token (= arg-max token id) times text_projection.  tokens int64 [B,77]."""
    p = prefix
    x = W[p + "token_embedding.weight"][tokens] + W[p + "positional_embedding"]
    L = x.shape[1]
    causal = torch.full((L, L), float("-inf")).triu_(1)                      # model.py:463-469
    x = x.permute(1, 0, 2)
    n_layers = 1 + max(int(k[len(p + "transformer.resblocks."):].split(".")[0]) for k in W
                       if k.startswith(p + "transformer.resblocks."))
    for i in range(n_layers):
        bp = f"{p}transformer.resblocks.{i}."
        h = _ln(x, W, bp + "ln_1")
        x = x + _mha(W, bp + "attn.", h, h, h, causal, heads)
        h = _ln(x, W, bp + "ln_2")
        h = F.linear(h, W[bp + "mlp.c_fc.weight"], W[bp + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, W[bp + "mlp.c_proj.weight"], W[bp + "mlp.c_proj.bias"])
    x = _ln(x.permute(1, 0, 2), W, p + "ln_final")
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ W[p + "text_projection"]


def clip_text_ensemble(tokens_per_template, W, prefix="clip_adapter.clip_model.", heads=8):
    """ClipAdapter.encode_text (adapter.py:121-138): per template L2-normalised embeddings, mean over templates,
    L2-normalise.  tokens_per_template int64 [n_templates, K, 77] -> [K, E]."""
    bucket = []
    for tok in tokens_per_template:
        e = clip_encode_text(tok, W, prefix, heads)
        bucket.append(e / e.norm(dim=-1, keepdim=True))
    e = torch.stack(bucket).mean(dim=0)
    return e / e.norm(dim=-1, keepdim=True)


def clip_tower_input(regions, resolution=224):
    """adapter.py:141-142: `/255`, bicubic resize to the tower's resolution (identity on roi_align's output), CLIP normalisation."""
    image = F.interpolate(regions / 255., (resolution, resolution), mode="bicubic")
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (image - mean) / std


def clip_encode_image(regions, W, prefix="clip_adapter.clip_model.visual.", resolution=224, heads=12, mask_regions=None,
                      mask_prompt_depth=0):
    """adapter.py:140-144; with mask_regions: AdaptedClipAdapter.encode_image (mask_adapted_adapter.py:143-147)."""
    feat = clip_visual(clip_tower_input(regions, resolution), W, prefix, heads, mask_regions, mask_prompt_depth)
    return feat / feat.norm(dim=-1, keepdim=True)


# ----------------------------------------------------------------------------------------------
# A12 + OpenVIS.open_vocabulary_inference — openvis/openvis.py:110-147 ; A16 inference_video — video_maskformer.py:262-298
# ----------------------------------------------------------------------------------------------
def open_vocabulary_inference(masks, frames, text_features, W, part_len=5, temperature=100.0, clip_heads=12,
                              clip_resolution=224, mask_prompt_depth=None, mask_prompt_fwd=True):
    """masks [Q,T,Hp,Wp] logits (already upsampled); frames [T,3,H,W] uint8; text_features [K,512] unit rows.
    mask_prompt_depth None: ClipAdapter (adapter.py:56-71); an int: AdaptedClipAdapter (mask_adapted_adapter.py:58-77),
    whose tower sees the mask regions when mask_prompt_fwd.
    Returns (probs [Qv,K], masks[valid_query], extras)."""
    T = frames.shape[0]
    clip_cls, valid_flag, boxes, embeds = [], [], [], []
    for idx in range(0, T, part_len):
        part_frames = frames[idx:idx + part_len]
        part_masks = masks[:, idx:idx + part_len].sigmoid().transpose(0, 1).contiguous()
        regions, valid, sb, mregions = clip_crops(part_frames, part_masks, clip_resolution, return_mask_regions=True)
        if regions is None:
            logits = torch.empty(0, text_features.shape[0])
        else:
            if mask_prompt_depth is not None and mask_prompt_fwd:
                feat = clip_encode_image(regions, W, resolution=clip_resolution, heads=clip_heads, mask_regions=mregions,
                                         mask_prompt_depth=mask_prompt_depth)
            else:
                feat = clip_encode_image(regions, W, resolution=clip_resolution, heads=clip_heads)
            logits = temperature * feat @ text_features.T                     # adapter.py:146-147
            boxes.append(sb)
            embeds.append(feat)
        clip_cls.append(logits)
        valid_flag.append(valid)
    clip_cls = torch.cat(clip_cls)
    valid_flag = torch.cat(valid_flag)
    probs, vmasks, mean_cls = aggregate_crop_logits(clip_cls, valid_flag, masks)
    if mean_cls is None:
        return [], [], {}
    extras = {"crop_logits": clip_cls, "valid": valid_flag, "boxes": torch.cat(boxes) if boxes else None,
              "query_logits": mean_cls, "crop_embeds": torch.cat(embeds) if embeds else None}     # unit rows [M, E]: what `feat` is at adapter.py:144-145
    return probs, vmasks, extras


def aggregate_crop_logits(clip_cls, valid_flag, masks):
    """openvis.py:126-142: crop logits [M,K] in (frame, query) order of `valid_flag` [T,Q] -> per-query mean over the frames with a
    crop -> softmax over K; the masks of the queries with at least one crop.  ([], [], None) when no crop is valid (:127-128)."""
    if torch.sum(valid_flag) == 0:
        return [], [], None
    valid_ids = torch.nonzero(valid_flag)
    valid_query_flag = torch.sum(valid_flag, dim=0) > 0
    valid_query_ids = torch.nonzero(valid_query_flag)[:, 0]
    query_clip_cls = [torch.mean(clip_cls[valid_ids[:, 1] == q], dim=0) for q in valid_query_ids]
    mean_cls = torch.stack(query_clip_cls)
    return mean_cls.softmax(dim=-1), masks[valid_query_flag], mean_cls


def inference_video(num_queries, num_classes, pred_cls, pred_masks, img_size, out_h, out_w, topk=10):
    """video_maskformer.py:262-298."""
    if len(pred_cls) == 0:
        return {"image_size": (out_h, out_w), "pred_entropys": [], "pred_scores": [], "pred_labels": [],
                "pred_masks": []}
    scores = pred_cls
    labels = torch.arange(num_classes).unsqueeze(0).repeat(num_queries, 1).flatten(0, 1)
    scores_per_image, topk_indices = scores.flatten(0, 1).topk(topk, sorted=False)
    labels_per_image = labels[topk_indices]
    topk_indices = topk_indices // num_classes
    entropys = torch.sum(-scores[topk_indices] * torch.log(scores[topk_indices]), dim=-1)
    pm = pred_masks[topk_indices]
    pm = pm[:, :, : img_size[0], : img_size[1]]
    pm = F.interpolate(pm, size=(out_h, out_w), mode="bilinear", align_corners=False)
    masks = pm > 0.
    return {"image_size": (out_h, out_w), "pred_entropys": entropys.tolist(), "pred_scores": scores_per_image.tolist(),
            "pred_labels": labels_per_image.tolist(), "pred_masks": [m for m in masks], "rows": topk_indices.tolist()}


def openvis_forward(frames, W, text_features, out_hw=None, stages=None, clip_heads=12, clip_resolution=224, backbone_fn=None,
                    mask_prompt_depth=None, mask_prompt_fwd=True):
    """OpenVIS.forward, eval (openvis/openvis.py:47-108). frames: uint8 [T,3,H,W].  mask_prompt_depth: see
    open_vocabulary_inference (None = ClipAdapter, int = AdaptedClipAdapter)."""
    T = frames.shape[0]
    images, (H, Wd) = preprocess([f for f in frames])
    feats = (backbone_fn or resnet50)(images, W)
    mask_features, _, ms = pixel_decoder(feats, W)
    cls, pred_masks = video_decoder(ms, mask_features, W)
    mask_pred = pred_masks[0]                                                # [Q,T,h,w]
    ih, iw = images.shape[-2:]
    mask_pred = F.interpolate(mask_pred, size=(ih, iw), mode="bilinear", align_corners=False)   # openvis.py:87-96
    probs, vmasks, extras = open_vocabulary_inference(mask_pred, frames, text_features, W, clip_heads=clip_heads,
                                                      clip_resolution=clip_resolution, mask_prompt_depth=mask_prompt_depth,
                                                      mask_prompt_fwd=mask_prompt_fwd)
    oh, ow = out_hw if out_hw is not None else (H, Wd)
    K = text_features.shape[0]
    out = inference_video(pred_masks.shape[1], K, probs, vmasks, (H, Wd), oh, ow)
    if stages is not None:
        stages.update(dict(images=images, feats=feats, mask_features=mask_features, ms=ms, pred_logits=cls,
                           pred_masks=pred_masks, probs=probs, **extras))
    return out


# ----------------------------------------------------------------------------------------------
# A7 (frame variant)  FrameMultiScaleMaskedTransformerDecoder — transformer_decoder/frame_mask2former_transformer_decoder.py:52-154
# ----------------------------------------------------------------------------------------------
def frame_prediction_heads(W, p, output, mask_features, target_size, nheads=8, with_class=True):
    """frame decoder:139-154. output [Q,bt,C]; mask_features [bt,C,H,W]."""
    dec = _ln(output, W, p + "decoder_norm").transpose(0, 1)
    outputs_class = F.linear(dec, W[p + "class_embed.weight"], W[p + "class_embed.bias"]) if with_class else None
    mask_embed = _mlp3(dec, W, p + "mask_embed.")
    outputs_mask = torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)
    am = F.interpolate(outputs_mask, size=target_size, mode="bilinear", align_corners=False)
    am = (am.sigmoid().flatten(2).unsqueeze(1).repeat(1, nheads, 1, 1).flatten(0, 1) < 0.5).bool()
    return outputs_class, outputs_mask, am


def frame_decoder(ms_feats, mask_features, W, prefix="sem_seg_head.predictor.", n_layers=9, nheads=8):
    """eval: bs = 1, every frame decoded independently. Returns dict with pred_logits [1,T,Q,C+1],
    pred_masks [1,Q,T,h,w], pred_embeds [1,T,Q,C]."""
    p = prefix
    src, pos, size_list = [], [], []
    for i in range(3):
        h, w = ms_feats[i].shape[-2:]
        size_list.append((h, w))
        pos.append(pe_sine_2d(ms_feats[i].shape[0], h, w).flatten(2).permute(2, 0, 1))
        src.append((ms_feats[i].flatten(2) + W[p + "level_embed.weight"][i][None, :, None]).permute(2, 0, 1))
    bs = src[0].shape[1]
    query_embed = W[p + "query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    output = W[p + "query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)
    cls, msk, attn_mask = frame_prediction_heads(W, p, output, mask_features, size_list[0], nheads)
    for i in range(n_layers):
        li = i % 3
        attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False
        cp = f"{p}transformer_cross_attention_layers.{i}."
        tgt2 = _mha(W, cp + "multihead_attn.", output + query_embed, src[li] + pos[li], src[li], attn_mask, nheads)
        output = _ln(output + tgt2, W, cp + "norm")
        sp = f"{p}transformer_self_attention_layers.{i}."
        qk = output + query_embed
        tgt2 = _mha(W, sp + "self_attn.", qk, qk, output, None, nheads)
        output = _ln(output + tgt2, W, sp + "norm")
        fp = f"{p}transformer_ffn_layers.{i}."
        tgt2 = F.linear(F.relu(F.linear(output, W[fp + "linear1.weight"], W[fp + "linear1.bias"])),
                        W[fp + "linear2.weight"], W[fp + "linear2.bias"])
        output = _ln(output + tgt2, W, fp + "norm")
        cls, msk, attn_mask = frame_prediction_heads(W, p, output, mask_features, size_list[(i + 1) % 3], nheads)
    pred_embeds = _ln(output, W, p + "decoder_norm")                                  # [Q, T, C]
    return {"pred_logits": cls.unsqueeze(0), "pred_masks": msk.permute(1, 0, 2, 3).unsqueeze(0),
            "pred_embeds": pred_embeds.permute(1, 0, 2).unsqueeze(0)}


# ----------------------------------------------------------------------------------------------
# A14  MinVIS tracker — openvis/modeling/minvis.py:28-72, 320-338 (scipy.optimize.linear_sum_assignment)
# ----------------------------------------------------------------------------------------------
def match_via_embeds(tgt_embeds, cur_embeds):
    from scipy.optimize import linear_sum_assignment
    cur_embeds = cur_embeds / cur_embeds.norm(dim=1)[:, None]
    tgt_embeds = tgt_embeds / tgt_embeds.norm(dim=1)[:, None]
    cos_sim = torch.mm(cur_embeds, tgt_embeds.transpose(0, 1))
    C = 1.0 * (1 - cos_sim)
    indices = linear_sum_assignment(C.transpose(0, 1))[1]
    return indices.tolist()


def video_match_via_embeds(embeds):
    """embeds [T,Q,C] -> (indices [T,Q] long, permuted embeds [T,Q,C])."""
    last = embeds[0]
    idx_list, out = [], []
    for i in range(embeds.shape[0]):
        indices = match_via_embeds(last, embeds[i])
        last = embeds[i][indices]
        idx_list.append(indices)
        out.append(last)
    return torch.tensor(idx_list), torch.stack(out)


def minvis_post_processing(outputs):
    """minvis.py:320-338 for bs = 1: reorder per-frame logits / masks by the tracker's indices."""
    idx, _ = video_match_via_embeds(outputs["pred_embeds"][0])
    T = idx.shape[0]
    ar = torch.arange(T)[:, None]
    out = dict(outputs)
    out["pred_logits"] = outputs["pred_logits"][0][ar, idx].unsqueeze(0)                                  # [1,T,Q,C]
    out["pred_masks"] = outputs["pred_masks"][0].transpose(0, 1)[ar, idx].transpose(0, 1).unsqueeze(0)    # [1,Q,T,h,w]
    out["indices"] = idx
    return out


def run_window_inference(images, backbone_fn, head_fn, window_size=30, clip_feats=None):
    """MinVIS.run_window_inference (minvis.py:340-362) and, with `clip_feats`, SAN.run_window_inference (san.py:285-307):
    ceil(T / window_size) windows of consecutive frames go through backbone + head one after the other (a ragged last
    window is whatever the slice returns); the per-frame outputs are concatenated along their time axis -- pred_logits /
    pred_embeds / class_attn_biases dim 1, pred_masks dim 2 (moved to the CPU as f32 by the reference, :358 / :305) --
    and the auxiliary outputs are dropped.  BriVIS' own copy (brivis.py:267-316) is broken in the reference
    and is not restated."""
    T = len(images)
    iters = T // window_size + (1 if T % window_size != 0 else 0)
    out_list = []
    for i in range(iters):
        b0, b1 = i * window_size, (i + 1) * window_size
        feats = backbone_fn(images[b0:b1])
        out = head_fn(feats) if clip_feats is None else head_fn(feats, [x[b0:b1] for x in clip_feats])
        out_list.append(out)
    outputs = {}
    for key, dim in (("pred_logits", 1), ("class_attn_biases", 1), ("pred_embeds", 1), ("pred_masks", 2)):
        if key in out_list[0]:
            outputs[key] = torch.cat([o[key] for o in out_list], dim=dim)
    outputs["pred_masks"] = outputs["pred_masks"].to(torch.float32)
    return outputs


def openvis_online_forward(frames, W, text_features, out_hw=None, stages=None, clip_heads=12, clip_resolution=224):
    """OpenVISOnline.forward, eval (openvis/openvis.py:176-281); part_len = 10."""
    images, (H, Wd) = preprocess([f for f in frames])
    feats = resnet50(images, W)
    mask_features, _, ms = pixel_decoder(feats, W)
    out = minvis_post_processing(frame_decoder(ms, mask_features, W))
    mask_pred = out["pred_masks"][0]
    ih, iw = images.shape[-2:]
    mask_pred = F.interpolate(mask_pred, size=(ih, iw), mode="bilinear", align_corners=False)
    probs, vmasks, extras = open_vocabulary_inference(mask_pred, frames, text_features, W, part_len=10,
                                                      clip_heads=clip_heads, clip_resolution=clip_resolution)
    oh, ow = out_hw if out_hw is not None else (H, Wd)
    res = inference_video(out["pred_masks"].shape[1], text_features.shape[0], probs, vmasks, (H, Wd), oh, ow)
    if stages is not None:
        stages.update(dict(images=images, feats=feats, pred_masks=out["pred_masks"], pred_embeds=out["pred_embeds"],
                           indices=out["indices"], probs=probs, **extras))
    return res


# ----------------------------------------------------------------------------------------------
# A13  SideAdapter — openvis/modeling/clip_adapter/side_adapter.py:147-270 (state-dict prefix clip_adapter.)
# ----------------------------------------------------------------------------------------------
def _clip_block(x, W, bp, heads, attn_mask=None):
    """BiasedResidualAttentionBlock.forward (side_adapter.py:70-78; blocks model.py:238-268), x [L,N,C]."""
    h = _ln(x, W, bp + "ln_1")
    x = x + _mha(W, bp + "attn.", h, h, h, attn_mask, heads)
    h = _ln(x, W, bp + "ln_2")
    h = F.linear(h, W[bp + "mlp.c_fc.weight"], W[bp + "mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, W[bp + "mlp.c_proj.weight"], W[bp + "mlp.c_proj.bias"])


def san_front_encode_image(x, W, prefix="clip_adapter.", broken_idx=9, merge_ids=(3, 6, 9), resolution=224):
    """side_adapter.py:147-174. x: raw padded frames [T,3,Hp,Wp] (0..255 float). Returns (mg_feats[3] [T,256,g,g],
    (cls [1,T,C], pix [T,C,g,g]))."""
    v = prefix + "clip_model.visual."
    heads = W[v + "proj"].shape[0] // 64
    x = F.interpolate(x / 255., (resolution, resolution), mode="bicubic")
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    patch = W[v + "conv1.weight"].shape[-1]
    x = F.conv2d(x, W[v + "conv1.weight"], None, stride=patch)
    b, _, h, w = x.shape
    x = x.reshape(b, x.shape[1], -1).permute(0, 2, 1).contiguous()
    cls = W[v + "class_embedding"] + torch.zeros(b, 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1) + W[v + "positional_embedding"]       # grid == pos-embed grid: resize is the identity
    x = _ln(x, W, v + "ln_pre").permute(1, 0, 2).contiguous()
    outs = [(x[0:1], x[1:].permute(1, 2, 0).reshape(b, -1, h, w).contiguous())]
    for i in range(broken_idx):
        x = _clip_block(x, W, f"{v}transformer.resblocks.{i}.", heads)
        outs.append((x[0:1], x[1:].permute(1, 2, 0).reshape(b, -1, h, w).contiguous()))
    mg = [f[1] for i, f in enumerate(outs) if i in merge_ids]
    mg = [F.conv2d(f, W[f"{prefix}attn_projs.{i}.weight"], W[f"{prefix}attn_projs.{i}.bias"]) for i, f in enumerate(mg)]
    return mg, outs[-1]


def san_build_attn_bias(attn_bias, num_heads, target_shape):
    """side_adapter.py:237-270 for one bias tensor [B,n,Q,H,W] -> [B*num_heads, Q+1+L, Q+1+L]."""
    b, num_head, num_sos, h, w = attn_bias.shape
    ab = F.adaptive_max_pool2d(attn_bias.reshape(b, num_head * num_sos, h, w), output_size=target_shape)
    ab = ab.reshape(b, num_head, num_sos, *target_shape)
    if num_head == 1:
        ab = ab.repeat(1, num_heads, 1, 1, 1)
    ab = ab.reshape(b * num_heads, num_sos, -1)
    L = ab.shape[-1]
    nb = ab.new_zeros(num_sos + 1 + L, num_sos + 1 + L)
    nb[:, :num_sos] = -100
    nb[:num_sos, num_sos] = -100
    nb[torch.arange(num_sos), torch.arange(num_sos)] = 0
    nb = nb[None, ...].expand(b * num_heads, -1, -1).clone()
    nb[..., :num_sos, -L:] = ab
    return nb


def san_post_encode_image(feats, attn_bias, W, prefix="clip_adapter.", broken_idx=9, num_sos=100):
    """side_adapter.py:176-209. feats = (cls [1,T,C], pix [T,C,g,g]); attn_bias [T,n,Q,H,W] -> sos feats [T,Q,E] (unit rows)."""
    v = prefix + "clip_model.visual."
    heads = W[v + "proj"].shape[0] // 64
    cls_token, pix = feats
    n, c, h, w = pix.shape
    x = torch.cat([cls_token, pix.reshape(n, c, -1).permute(2, 0, 1)])
    sos = cls_token.repeat(num_sos, 1, 1)
    n_layers = 1 + max(int(k[len(v + "transformer.resblocks."):].split(".")[0]) for k in W if k.startswith(v + "transformer.resblocks."))
    bias = san_build_attn_bias(attn_bias, heads, (h, w))
    x = torch.cat([sos, x], dim=0)
    for i in range(broken_idx, n_layers):
        x = _clip_block(x, W, f"{v}transformer.resblocks.{i}.", heads, bias)
    sos = x[:num_sos].permute(1, 0, 2)
    sos = _ln(sos, W, v + "ln_post") @ W[v + "proj"]
    return F.normalize(sos, dim=-1)


def san_text_with_bg(text_features, W, prefix="clip_adapter."):
    """encode_text(w_bg=True) tail (side_adapter.py:228-231): append the normalised learned background embedding."""
    return torch.cat([text_features, F.normalize(W[prefix + "bg_embed"], dim=-1)], dim=0)


def san_cal_sim_logits(text_feats, image_feats, W, prefix="clip_adapter."):
    return W[prefix + "clip_model.logit_scale"].exp() * image_feats @ text_feats.T      # side_adapter.py:234-235


# side-adapter frame decoder — transformer_decoder/side_adapter_frame_mask2former_transformer_decoder.py:57-169
def side_frame_decoder(ms_feats, mask_features, W, prefix="sem_seg_head.predictor.", n_layers=9, nheads=8, clip_heads=12):
    p = prefix
    bt, c = mask_features.shape[:2]
    af = F.interpolate(mask_features, scale_factor=0.25, mode="bilinear", align_corners=False)
    ha, wa = af.shape[-2:]
    for j in range(3):
        af = F.conv2d(af, W[f"{p}attn_mlp.layers.{j}.weight"], W[f"{p}attn_mlp.layers.{j}.bias"])
        if j < 2:
            af = F.relu(af)
    af = af.reshape(bt, clip_heads, c, ha, wa)
    src, pos, size_list = [], [], []
    for i in range(3):
        h, w = ms_feats[i].shape[-2:]
        size_list.append((h, w))
        pos.append(pe_sine_2d(ms_feats[i].shape[0], h, w).flatten(2).permute(2, 0, 1))
        src.append((ms_feats[i].flatten(2) + W[p + "level_embed.weight"][i][None, :, None]).permute(2, 0, 1))
    bs = src[0].shape[1]
    query_embed = W[p + "query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    output = W[p + "query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)

    def heads(out, target):
        dec = _ln(out, W, p + "decoder_norm").transpose(0, 1)
        attn_embed = _mlp3(dec, W, p + "attn_embed.")
        mask_embed = _mlp3(dec, W, p + "mask_embed.")
        biases = torch.einsum("bqc,bnchw->bnqhw", attn_embed, af)
        masks = torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)
        am = F.interpolate(masks, size=target, mode="bilinear", align_corners=False)
        am = (am.sigmoid().flatten(2).unsqueeze(1).repeat(1, nheads, 1, 1).flatten(0, 1) < 0.5).bool()
        return biases, masks, am

    biases, msk, attn_mask = heads(output, size_list[0])
    for i in range(n_layers):
        li = i % 3
        attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False
        cp = f"{p}transformer_cross_attention_layers.{i}."
        tgt2 = _mha(W, cp + "multihead_attn.", output + query_embed, src[li] + pos[li], src[li], attn_mask, nheads)
        output = _ln(output + tgt2, W, cp + "norm")
        sp = f"{p}transformer_self_attention_layers.{i}."
        qk = output + query_embed
        output = _ln(output + _mha(W, sp + "self_attn.", qk, qk, output, None, nheads), W, sp + "norm")
        fp = f"{p}transformer_ffn_layers.{i}."
        tgt2 = F.linear(F.relu(F.linear(output, W[fp + "linear1.weight"], W[fp + "linear1.bias"])),
                        W[fp + "linear2.weight"], W[fp + "linear2.bias"])
        output = _ln(output + tgt2, W, fp + "norm")
        biases, msk, attn_mask = heads(output, size_list[(i + 1) % 3])
    pred_embeds = _ln(output, W, p + "decoder_norm")
    return {"class_attn_biases": biases.unsqueeze(0), "pred_masks": msk.permute(1, 0, 2, 3).unsqueeze(0),
            "pred_embeds": pred_embeds.permute(1, 0, 2).unsqueeze(0), "attn_feats": af, "mask_feats": mask_features}


def side_video_decoder(ms_feats, mask_features, W, prefix="sem_seg_head.predictor.", n_layers=9, nheads=8, clip_heads=12):
    """SideAdapterVideoMultiScaleMaskedTransformerDecoder.forward, eval: bs = 1, t = T
    (side_adapter_video_mask2former_transformer_decoder.py:51-142)."""
    p = prefix
    bt, c, h_m, w_m = mask_features.shape
    bs, t = 1, bt
    af = F.interpolate(mask_features, scale_factor=0.25, mode="bilinear", align_corners=False)
    ha, wa = af.shape[-2:]
    for j in range(3):
        af = F.conv2d(af, W[f"{p}attn_mlp.layers.{j}.weight"], W[f"{p}attn_mlp.layers.{j}.bias"])
        if j < 2:
            af = F.relu(af)
    af = af.reshape(bs, t, clip_heads, c, ha, wa)
    mf = mask_features.view(bs, t, c, h_m, w_m)
    src, pos, size_list = [], [], []
    for i in range(3):
        h, w = ms_feats[i].shape[-2:]
        size_list.append((h, w))
        pe = pe_sine_3d(bs, t, h, w).flatten(3)
        s = ms_feats[i].flatten(2) + W[p + "level_embed.weight"][i][None, :, None]
        pos.append(pe.view(bs, t, c, h * w).permute(1, 3, 0, 2).flatten(0, 1))
        src.append(s.view(bs, t, c, h * w).permute(1, 3, 0, 2).flatten(0, 1))
    query_embed = W[p + "query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    output = W[p + "query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)

    def heads(out, target):
        dec = _ln(out, W, p + "decoder_norm").transpose(0, 1)
        attn_embed = _mlp3(dec, W, p + "attn_embed.")
        mask_embed = _mlp3(dec, W, p + "mask_embed.")
        biases = torch.einsum("bqc,btnchw->btnqhw", attn_embed, af)
        masks = torch.einsum("bqc,btchw->bqthw", mask_embed, mf)
        b, q, tt = masks.shape[:3]
        am = F.interpolate(masks.flatten(0, 1), size=target, mode="bilinear", align_corners=False).view(b, q, tt, *target)
        am = (am.sigmoid().flatten(2).unsqueeze(1).repeat(1, nheads, 1, 1).flatten(0, 1) < 0.5).bool()
        return biases, masks, am

    biases, msk, attn_mask = heads(output, size_list[0])
    for i in range(n_layers):
        li = i % 3
        attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False
        cp = f"{p}transformer_cross_attention_layers.{i}."
        tgt2 = _mha(W, cp + "multihead_attn.", output + query_embed, src[li] + pos[li], src[li], attn_mask, nheads)
        output = _ln(output + tgt2, W, cp + "norm")
        sp = f"{p}transformer_self_attention_layers.{i}."
        qk = output + query_embed
        output = _ln(output + _mha(W, sp + "self_attn.", qk, qk, output, None, nheads), W, sp + "norm")
        fp = f"{p}transformer_ffn_layers.{i}."
        tgt2 = F.linear(F.relu(F.linear(output, W[fp + "linear1.weight"], W[fp + "linear1.bias"])),
                        W[fp + "linear2.weight"], W[fp + "linear2.bias"])
        output = _ln(output + tgt2, W, fp + "norm")
        biases, msk, attn_mask = heads(output, size_list[(i + 1) % 3])
    return {"class_attn_biases": biases, "pred_masks": msk}


def san_forward(frames, W, text_features, out_hw=None, stages=None, broken_idx=9, merge_ids=(3, 6, 9), resolution=224,
                clip_heads=12, num_queries=100, backbone_fn=None):
    """SAN.forward, eval (san.py:85-144)."""
    images, (H, Wd) = preprocess([f for f in frames])
    Hp, Wp = images.shape[-2:]
    ori = torch.zeros(frames.shape[0], 3, Hp, Wp)
    ori[:, :, :H, :Wd] = frames.float()
    mg, bk = san_front_encode_image(ori, W, broken_idx=broken_idx, merge_ids=merge_ids, resolution=resolution)
    tf = san_text_with_bg(text_features, W)
    feats = (backbone_fn or resnet50)(images, W)
    mask_features, _, ms = pixel_decoder(feats, W, extra_features=mg)
    out = side_video_decoder(ms, mask_features, W, clip_heads=clip_heads)
    sos = san_post_encode_image(bk, out["class_attn_biases"][0], W, broken_idx=broken_idx, num_sos=num_queries)
    logits = san_cal_sim_logits(tf, sos, W)                                               # [T,Q,K+1]
    probs = F.softmax(logits.mean(dim=0), dim=-1)[:, :-1]                                 # san.py:116; postprocess :218-219
    mask_pred = F.interpolate(out["pred_masks"][0], size=(Hp, Wp), mode="bilinear", align_corners=False)
    oh, ow = out_hw if out_hw is not None else (H, Wd)
    res = inference_video(mask_pred.shape[0], text_features.shape[0], probs, mask_pred, (H, Wd), oh, ow)
    if stages is not None:
        stages.update(dict(pred_masks=out["pred_masks"], pred_logits=logits.unsqueeze(0), probs=probs,
                           class_attn_biases=out["class_attn_biases"]))
    return res


def san_online_image_outputs(frames, W, text_features, broken_idx=9, merge_ids=(3, 6, 9), resolution=224, clip_heads=12,
                             num_queries=100, backbone_fn=None):
    """SANOnline.forward up to the per-frame logits (san.py:211-231). frames uint8 [T,3,H,W]; text_features [K,E]."""
    images, (H, Wd) = preprocess([f for f in frames])
    Hp, Wp = images.shape[-2:]
    ori = torch.zeros(frames.shape[0], 3, Hp, Wp)
    ori[:, :, :H, :Wd] = frames.float()                                   # ImageList.from_tensors(ori_images): zero pad
    mg, bk = san_front_encode_image(ori, W, broken_idx=broken_idx, merge_ids=merge_ids, resolution=resolution)
    tf = san_text_with_bg(text_features, W)
    feats = (backbone_fn or resnet50)(images, W)
    mask_features, _, ms = pixel_decoder(feats, W, extra_features=mg)
    out = side_frame_decoder(ms, mask_features, W, clip_heads=clip_heads)
    sos = san_post_encode_image(bk, out["class_attn_biases"][0], W, broken_idx=broken_idx, num_sos=num_queries)
    out["pred_logits"] = san_cal_sim_logits(tf, sos, W).unsqueeze(0)       # [1,T,Q,K+1]
    out.update(images=images, image_size=(H, Wd), clip_bk=bk, text_feats=tf)
    return out


def temporal_mean_post_processing(pred_logits, pred_masks, image_hw, num_classes):
    """BriVIS.post_processing (brivis.py:242-265), the same lines inline in SANOnline.forward (san.py:257-273): pred_logits [1,T,Q,C] ->
    mean over T -> softmax and drop of the last (background) column when C == num_classes + 1; pred_masks [1,Q,T,h,w] -> bilinear to
    the padded image size.  Returns (probs [Q,K], masks [Q,T,Hp,Wp])."""
    cls = pred_logits.mean(dim=1)[0]
    mask_pred = pred_masks[0]
    if cls.shape[-1] == num_classes + 1:
        cls = F.softmax(cls, dim=-1)[:, :-1]
    if tuple(mask_pred.shape[-2:]) != tuple(image_hw):
        mask_pred = F.interpolate(mask_pred, size=tuple(image_hw), mode="bilinear", align_corners=False)
    return cls, mask_pred


def san_online_forward(frames, W, text_features, out_hw=None, stages=None, **kw):
    """SANOnline.forward, eval (san.py:177-283)."""
    out = san_online_image_outputs(frames, W, text_features, **kw)
    images, (H, Wd) = out["images"], out["image_size"]
    out = minvis_post_processing(out)
    probs, mask_pred = temporal_mean_post_processing(out["pred_logits"], out["pred_masks"], images.shape[-2:], text_features.shape[0])
    oh, ow = out_hw if out_hw is not None else (H, Wd)
    res = inference_video(mask_pred.shape[0], text_features.shape[0], probs, mask_pred, (H, Wd), oh, ow)
    if stages is not None:
        stages.update(dict(pred_masks=out["pred_masks"], pred_logits=out["pred_logits"], indices=out["indices"], probs=probs,
                           pred_embeds=out["pred_embeds"]))
    return res


# ----------------------------------------------------------------------------------------------
# A15  TemporalInstanceResampler — openvis/modeling/resampler.py:244-316 (state-dict prefix resampler.)
# ----------------------------------------------------------------------------------------------
def resampler_forward(frame_embeds, W, prefix="resampler.", n_layers=6, nheads=8):
    """frame_embeds [1,T,Q,C] (tracker order) -> temporal_tgt after the 6 layers as [T,Q,C] (before decode_norm).
    Only the last prediction head is consumed at eval (resampler.py:294-296), see brivis_forward."""
    p = prefix
    _, t, q, c = frame_embeds.shape
    x = frame_embeds[0]                                                   # 't (b q) c' with b = 1
    for i in range(n_layers):
        lp = f"{p}long_aggregate_layers.{i}."
        x = _ln(x + _mha(W, lp + "self_attn.", x, x, x, None, nheads), W, lp + "norm")     # attention over T per query
        s = x.permute(1, 2, 0)                                            # (q, c, t)
        sp = f"{p}short_aggregate_layers.{i}."
        y = F.conv1d(F.pad(s, (2, 2), mode="replicate"), W[sp + "0.weight"], W[sp + "0.bias"])
        y = F.conv1d(F.pad(F.relu(y), (1, 1), mode="replicate"), W[sp + "2.weight"], W[sp + "2.bias"])
        x = _ln((y + s).transpose(1, 2), W, f"{p}aggregate_norms.{i}").permute(1, 0, 2)
        fp = f"{p}transformer_ffn_layers.{i}."
        x = _ln(x + F.linear(F.relu(F.linear(x, W[fp + "linear1.weight"], W[fp + "linear1.bias"])),
                             W[fp + "linear2.weight"], W[fp + "linear2.bias"]), W, fp + "norm")
    return x


def resampler_heads(x, mask_feats, attn_feats, W, prefix="resampler."):
    """forward_prediction_heads (resampler.py:304-316) without the CLIP pass: x [T,Q,C] -> (masks [T,Q,h,w], biases [T,n,Q,ha,wa])."""
    out = _ln(x, W, prefix + "decode_norm")
    masks = torch.einsum("bqc,bchw->bqhw", _mlp3(out, W, prefix + "mask_embed."), mask_feats)
    biases = torch.einsum("bqc,bnchw->bnqhw", _mlp3(out, W, prefix + "attn_embed."), attn_feats)
    return masks, biases, out


def brivis_forward(frames, W, text_features, out_hw=None, stages=None, **kw):
    """BriVIS.forward, eval, WINDOW_INFERENCE False (brivis.py:131-176, 201-211, 242-265)."""
    num_queries = kw.get("num_queries", 100)
    io = san_online_image_outputs(frames, W, text_features, **kw)
    images, (H, Wd) = io["images"], io["image_size"]
    idx, frame_embeds = video_match_via_embeds(io["pred_embeds"][0])                     # brivis.py:173
    x = resampler_forward(frame_embeds.unsqueeze(0), W)
    masks, biases, emb = resampler_heads(x, io["mask_feats"], io["attn_feats"], W)
    sos = san_post_encode_image(io["clip_bk"], biases, W, broken_idx=kw.get("broken_idx", 9), num_sos=num_queries)
    logits = san_cal_sim_logits(io["text_feats"], sos, W)                                # [T,Q,K+1]
    pred_masks = masks.permute(1, 0, 2, 3)                                               # [Q,T,h,w]
    probs, mask_pred = temporal_mean_post_processing(logits.unsqueeze(0), pred_masks.unsqueeze(0), images.shape[-2:],
                                                     text_features.shape[0])              # brivis.py:242-265
    oh, ow = out_hw if out_hw is not None else (H, Wd)
    res = inference_video(num_queries, text_features.shape[0], probs, mask_pred, (H, Wd), oh, ow)
    if stages is not None:
        stages.update(dict(pred_masks=pred_masks.unsqueeze(0), pred_logits=logits.unsqueeze(0), indices=idx, probs=probs,
                           pred_embeds=emb))
    return res
