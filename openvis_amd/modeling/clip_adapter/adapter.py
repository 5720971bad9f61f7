"""ClipAdapter — mirror of openvis/modeling/clip_adapter/adapter.py:34-147 (eval path) on the gfx950 kernels.

forward(frames, text, masks) keeps the reference's meaning, but `masks` are the LOW-RESOLUTION mask logits
[Q,T,h,w] plus the padded size: the x4 upsample + sigmoid + threshold + bounding box + roi_align + blend +
CLIP normalisation of openvis.py:87-96,118 / adapter.py:73-116,140-143 are evaluated inside two kernels
(ovis_mask_bbox, ovis_clip_crop_patches) without materialising [Q,T,Hp,Wp].

dtype: `precision="fp16"` (default) rounds the ViT GEMM operands to fp16 like the reference's GPU path
(clip.load on cuda is fp16; crops are .half()'d, adapter.py:108-111) with f32 accumulation, residual stream,
LayerNorm and softmax; `precision="fp32"` runs the tower on the exact-f32 matrix cores (the oracle's dtype).
Text side (adapter.py:121-138): embeddings come from `text_cache` (filled by `set_text_features`); the CLIP text
tower / tokenizer is a later §8(f) row, so an unknown class name raises instead of being encoded."""
import numpy as np
import torch

from ... import ops

PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)
PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)

_CLIP_ARCH = {"ViT-B/16": dict(width=768, layers=12, heads=12, patch=16, resolution=224, embed_dim=512),
              "ViT-L/14@336px": dict(width=1024, layers=24, heads=16, patch=14, resolution=336, embed_dim=768)}


class ClipVisual:
    """CLIP VisionTransformer.forward (mask_adapted_clip/model.py:327-362): m=None, or the mask-prompt path when
    `patch_open` (the pooled, ceil'ed mask of model.py:332-333 as 0/1 bytes) is given."""

    def __init__(self, width, layers, heads, patch, resolution, embed_dim, precision="fp16", mask_prompt_depth=0):
        self.width, self.layers, self.heads, self.patch = width, layers, heads, patch
        self.mask_prompt_depth = mask_prompt_depth
        self.input_resolution, self.output_dim = resolution, embed_dim
        if precision not in ("fp16", "fp32"):
            raise ValueError("precision must be 'fp16' (GEMM operands fp16 like the reference's GPU CLIP) or 'fp32'")
        self.precision = precision
        # fp16 policy only: keep the residual stream of plain (no mask prompt, no attention bias) forward passes in fp16, as the
        # reference's fp16 CLIP does (adapter.py:108-111); set by build_clip_adapter from MODEL.CLIP_ADAPTER.RESIDUAL_STREAM
        self.stream16 = False
        # fp16 residual stream only: ln_1 / ln_2 folded into in_proj / c_fc (ops.fold_layernorm; the normalised rows are never
        # written, include/openvis_hip.h ovis_gemm_nt_f16_ln).  False: LayerNorm kernel + plain GEMM (the round-2 path).
        self.fold_ln = True
        self.w = {}

    def load_state_dict(self, sd, prefix, device):
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        w = self.w = {}                                             # (a second load must not meet the first one's folded operands: _fold caches in w)
        cw = g("conv1.weight")
        cw = cw.view(cw.shape[0], -1)
        w["conv1"] = torch.nn.functional.pad(cw, (0, ops.patch_row_len(self.patch) - cw.shape[1])).contiguous()   # zero pad columns
        w["cls"], w["pos"] = g("class_embedding"), g("positional_embedding")
        for n in ("ln_pre", "ln_post"):
            w[n + ".w"], w[n + ".b"] = g(n + ".weight"), g(n + ".bias")
        w["proj_t"] = g("proj").t().contiguous()                   # x @ proj == gemm_nt(x, proj^T)
        if self.mask_prompt_depth > 0:
            w["mask_embedding"] = g("mask_embedding")              # [depth, G*G or 1, width] (model.py:325)
        for i in range(self.layers):
            p = f"transformer.resblocks.{i}."
            for k in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                      "ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                      "mlp.c_proj.weight", "mlp.c_proj.bias"):
                w[f"{i}.{k}"] = g(p + k)
        if self.precision == "fp16":                                  # GEMM operand copies in fp16 (cast once)
            for k in ["conv1"] + [f"{i}.{n}" for i in range(self.layers)
                                  for n in ("attn.in_proj_weight", "attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")]:
                w[k + ".h"] = ops.cast_f16(w[k])
        return self

    def _fold(self, i, which):
        """LayerNorm-folded operands (wg fp16, s, c) of block i: "qkv" = ln_1 -> in_proj, "fc" = ln_2 -> c_fc, "kv" = ln_1 -> the key | value
        rows of in_proj (last_block_cls).  Built at first use (only a tower that runs the folded fp16 stream pays for them: ~350 MB for
        ViT-L/14) and cached; two ClipPipeline threads racing here build the same values."""
        key = f"{i}.{which}.fold"
        w = self.w
        if key not in w:
            C = self.width
            g1, b1 = w[f"{i}.ln_1.weight"], w[f"{i}.ln_1.bias"]
            if which == "qkv":
                w[key] = ops.fold_layernorm(w[f"{i}.attn.in_proj_weight"], w[f"{i}.attn.in_proj_bias"], g1, b1)
            elif which == "fc":
                w[key] = ops.fold_layernorm(w[f"{i}.mlp.c_fc.weight"], w[f"{i}.mlp.c_fc.bias"], w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"])
            else:
                w[key] = ops.fold_layernorm(w[f"{i}.attn.in_proj_weight"][C:].contiguous(), w[f"{i}.attn.in_proj_bias"][C:].contiguous(), g1, b1)
        return w[key]

    def embed(self, A, M, patch_open=None):
        """patch im2col matrix [M*G*G, 3*ps*ps] (f32 or fp16) -> ln_pre(tokens) [M, G*G+1, C] (model.py:328-343): f32, or fp16 when
        the tower keeps an fp16 residual stream (stream16, plain crops only)."""
        w = self.w
        G = self.input_resolution // self.patch
        s16 = self.stream16 and self.precision == "fp16" and patch_open is None
        x = ops.gemm_nt_f16(A, w["conv1.h"], out_f16=s16) if self.precision == "fp16" else ops.gemm_nt(A, w["conv1"], cw=True)   # conv1, no bias
        if patch_open is not None:                                  # model.py:334-338
            ops.mask_prompt_select(x, patch_open, w["mask_embedding"][0], 0)
        return ops.vit_embed_ln(x, w["cls"], w["pos"], w["ln_pre.w"], w["ln_pre.b"], M, G * G + 1)

    def run_blocks(self, x, i0, i1, attn_bias=None, stats=None, with_stats=False):
        """resblocks[i0:i1] on x [B, L, C] (model.py:238-268); attn_bias: additive f32 [B, heads, L, ld] (SideAdapter).
        stats: (mean, rstd) f32 [B*L, 2] of the rows of x if the GEMM that wrote x left them (the folded fp16 stream), else None;
        with_stats=True returns (x, stats of the rows of the returned x or None) -- the hand-over is explicit, the caller drops the statistics
        when it rewrites rows in place (forward_patches' mask prompt).
        fp16 policy: GEMM operands rounded to fp16 (weights cast once), f32 accumulation / LayerNorm statistics / softmax; the
        residual stream is whatever dtype x has: f32, or fp16 (stream16: x_new = fp16(f32(x) + bias + sum), one rounding per
        sub-block -- what the reference's fp16 CLIP keeps between blocks).  fp32 policy: exact-f32 MFMA, f32 stream."""
        w = self.w
        B, L, C = x.shape
        Hh = self.heads
        D = C // Hh
        f16 = self.precision == "fp16"
        if f16 and attn_bias is None and x.dtype == torch.float16:
            x, st = self._run_blocks_stream16(x, i0, i1, stats)
            return (x, st) if with_stats else x
        for i in range(i0, i1):
            h = ops.layernorm(x, w[f"{i}.ln_1.weight"], w[f"{i}.ln_1.bias"], out_f16=f16)
            if f16 and attn_bias is None:
                qkv = ops.gemm_nt_f16(h.view(-1, C), w[f"{i}.attn.in_proj_weight.h"], w[f"{i}.attn.in_proj_bias"], out_f16=True)
                att = ops.attention_f16(qkv, qkv[:, C:], qkv[:, 2 * C:], B, Hh, L, L, D, L * 3 * C, 3 * C, L * 3 * C, 3 * C,
                                        L * 3 * C, 3 * C)
            else:
                if f16:
                    qkv = ops.gemm_nt_f16(h.view(-1, C), w[f"{i}.attn.in_proj_weight.h"], w[f"{i}.attn.in_proj_bias"])
                else:
                    qkv = ops.gemm_nt(h.view(-1, C), w[f"{i}.attn.in_proj_weight"], w[f"{i}.attn.in_proj_bias"], cw=True)
                att = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B, Hh, L, L, D, L * 3 * C, 3 * C, L * 3 * C, 3 * C,
                                    L * 3 * C, 3 * C, bias=attn_bias, out_f16=f16)
            if f16:
                x = ops.gemm_nt_f16(att.view(-1, C), w[f"{i}.attn.out_proj.weight.h"], w[f"{i}.attn.out_proj.bias"],
                                    x.view(-1, C)).view(B, L, C)
                h = ops.layernorm(x, w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"], out_f16=True)
                f = ops.gemm_nt_f16(h.view(-1, C), w[f"{i}.mlp.c_fc.weight.h"], w[f"{i}.mlp.c_fc.bias"], None,
                                    ops.ACT_QUICKGELU, out_f16=True)
                x = ops.gemm_nt_f16(f, w[f"{i}.mlp.c_proj.weight.h"], w[f"{i}.mlp.c_proj.bias"], x.view(-1, C)).view(B, L, C)
            else:
                x = ops.gemm_nt(att.view(-1, C), w[f"{i}.attn.out_proj.weight"], w[f"{i}.attn.out_proj.bias"],
                                x.view(-1, C), cw=True).view(B, L, C)
                h = ops.layernorm(x, w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"])
                f = ops.gemm_nt(h.view(-1, C), w[f"{i}.mlp.c_fc.weight"], w[f"{i}.mlp.c_fc.bias"], None, ops.ACT_QUICKGELU, cw=True)
                x = ops.gemm_nt(f, w[f"{i}.mlp.c_proj.weight"], w[f"{i}.mlp.c_proj.bias"], x.view(-1, C), cw=True).view(B, L, C)
        return (x, None) if with_stats else x

    def _run_blocks_stream16(self, x, i0, i1, stats=None):
        """run_blocks on the fp16 residual stream (plain blocks).  Where the ping-pong kernel takes the problem, ln_1 / ln_2 are folded
        into in_proj / c_fc (ops.gemm_nt_f16_ln: the normalised rows are never written) and the row statistics come out of the epilogue
        of the GEMM that wrote the stream (ops.gemm_nt_f16_res16_stats); otherwise LayerNorm kernel + plain GEMM, per GEMM."""
        w = self.w
        B, L, C = x.shape
        Hh = self.heads
        D = C // Hh
        M = B * L
        Q = ops.ACT_QUICKGELU
        can = self.fold_ln and i1 > i0
        fold_qkv = can and ops.gemm_nt_f16_ln_eligible(M, 3 * C, C)
        fold_fc = can and ops.gemm_nt_f16_ln_eligible(M, w[f"{i0}.mlp.c_fc.weight"].shape[0], C, Q)
        st = stats if fold_qkv else None                              # (mean, rstd) of the rows of x, if the GEMM that wrote x left them
        x = x.view(M, C)
        for i in range(i0, i1):
            if fold_qkv:
                qkv = ops.gemm_nt_f16_ln(x, *self._fold(i, "qkv"), st if st is not None else ops.row_stats_f16(x))
            else:
                h = ops.layernorm(x, w[f"{i}.ln_1.weight"], w[f"{i}.ln_1.bias"], out_f16=True)
                qkv = ops.gemm_nt_f16(h, w[f"{i}.attn.in_proj_weight.h"], w[f"{i}.attn.in_proj_bias"], out_f16=True)
            att = ops.attention_f16(qkv, qkv[:, C:], qkv[:, 2 * C:], B, Hh, L, L, D, L * 3 * C, 3 * C, L * 3 * C, 3 * C, L * 3 * C, 3 * C)
            wo, bo = w[f"{i}.attn.out_proj.weight.h"], w[f"{i}.attn.out_proj.bias"]
            if fold_fc:
                x, st = ops.gemm_nt_f16_res16_stats(att.view(M, C), wo, bo, x)
                f = ops.gemm_nt_f16_ln(x, *self._fold(i, "fc"), st if st is not None else ops.row_stats_f16(x), Q)
            else:
                x = ops.gemm_nt_f16(att.view(M, C), wo, bo, x)
                h = ops.layernorm(x, w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"], out_f16=True)
                f = ops.gemm_nt_f16(h, w[f"{i}.mlp.c_fc.weight.h"], w[f"{i}.mlp.c_fc.bias"], None, Q, out_f16=True)
            wp, bp = w[f"{i}.mlp.c_proj.weight.h"], w[f"{i}.mlp.c_proj.bias"]
            if fold_qkv:                                              # the next block's ln_1 (or last_block_cls) reads these statistics
                x, st = ops.gemm_nt_f16_res16_stats(f, wp, bp, x)
            else:
                x, st = ops.gemm_nt_f16(f, wp, bp, x), None
        return x.view(B, L, C), (st if fold_qkv else None)            # no state on the model: clips in flight share the tower

    def last_block_cls(self, x, i, stats=None):
        """resblock i evaluated for the CLASS TOKEN only -> f32 [B, C].  ln_post reads x[:, 0] alone (model.py:356-358), so in
        the last block every token still contributes its key / value, but the query projection, the attention rows, the
        output projection and the whole MLP are needed for one token per crop: [B] rows instead of [B*L] (the same
        arithmetic per element as run_blocks, so the result matches it to the last bit of the GEMM's accumulation order).
        stats: the row statistics run_blocks(with_stats=True) handed over for THIS x (never after an in-place edit of x), or None."""
        w = self.w
        B, L, C = x.shape
        Hh = self.heads
        D = C // Hh
        f16 = self.precision == "fp16"
        fold = (f16 and self.fold_ln and x.dtype == torch.float16 and ops.gemm_nt_f16_ln_eligible(B * L, 2 * C, C))
        if fold:                                                     # keys | values from the raw rows; ln_1 itself only for the class rows
            h = None
            hq = ops.layernorm(x[:, 0, :].contiguous(), w[f"{i}.ln_1.weight"], w[f"{i}.ln_1.bias"], out_f16=True)
        else:
            h = ops.layernorm(x, w[f"{i}.ln_1.weight"], w[f"{i}.ln_1.bias"], out_f16=f16)
            hq = h[:, 0, :].contiguous()
        bi = w[f"{i}.attn.in_proj_bias"]
        # the class-token path is [B] rows: f32 from here on whatever the stream's dtype
        xc = ops.cast_f16_to_f32_rows(x[:, 0, :]) if x.dtype == torch.float16 else x[:, 0, :].contiguous()
        if f16:
            wi = w[f"{i}.attn.in_proj_weight.h"]
            if fold:
                kv = ops.gemm_nt_f16_ln(x.view(-1, C), *self._fold(i, "kv"), stats if stats is not None else ops.row_stats_f16(x))
            else:
                kv = ops.gemm_nt_f16(h.view(-1, C), wi[C:], bi[C:], out_f16=True)               # [B*L, 2C]: keys | values
            q = ops.gemm_nt_f16(hq, wi[:C], bi[:C], out_f16=True)                               # [B, C]
            att = ops.attention_f16(q, kv, kv[:, C:], B, Hh, 1, L, D, C, C, L * 2 * C, 2 * C, L * 2 * C, 2 * C)
            xc = ops.gemm_nt_f16(att.view(B, C), w[f"{i}.attn.out_proj.weight.h"], w[f"{i}.attn.out_proj.bias"], xc)
            h2 = ops.layernorm(xc, w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"], out_f16=True)
            f = ops.gemm_nt_f16(h2, w[f"{i}.mlp.c_fc.weight.h"], w[f"{i}.mlp.c_fc.bias"], None, ops.ACT_QUICKGELU, out_f16=True)
            return ops.gemm_nt_f16(f, w[f"{i}.mlp.c_proj.weight.h"], w[f"{i}.mlp.c_proj.bias"], xc)
        wi = w[f"{i}.attn.in_proj_weight"]
        kv = ops.gemm_nt(h.view(-1, C), wi[C:], bi[C:], cw=True)
        q = ops.gemm_nt(hq, wi[:C], bi[:C])
        att = ops.attention(q, kv, kv[:, C:], B, Hh, 1, L, D, C, C, L * 2 * C, 2 * C, L * 2 * C, 2 * C)
        xc = ops.gemm_nt(att.view(B, C), w[f"{i}.attn.out_proj.weight"], w[f"{i}.attn.out_proj.bias"], xc)
        h2 = ops.layernorm(xc, w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"])
        f = ops.gemm_nt(h2, w[f"{i}.mlp.c_fc.weight"], w[f"{i}.mlp.c_fc.bias"], None, ops.ACT_QUICKGELU)
        return ops.gemm_nt(f, w[f"{i}.mlp.c_proj.weight"], w[f"{i}.mlp.c_proj.bias"], xc)

    def head(self, tok):
        """ln_post + projection of token rows [N, C] -> [N, embed_dim] (model.py:358-361), exact f32."""
        w = self.w
        return ops.gemm_nt(ops.layernorm(tok, w["ln_post.w"], w["ln_post.b"]), w["proj_t"])

    def forward_patches(self, A, M, patch_open=None):
        """A: patch im2col matrix [M*G*G, 3*ps*ps] -> image features [M, embed_dim] (before L2 normalisation).
        patch_open uint8 [M, G*G]: the mask-prompt path (model.py:344-352) — after block d < mask_prompt_depth the
        closed patch tokens are reset to mask_embedding[d]."""
        last = self.layers - 1
        st = None
        if patch_open is None:
            x, st = self.run_blocks(self.embed(A, M), 0, last, with_stats=True)
        else:
            if self.mask_prompt_depth < 1:
                raise ValueError("mask prompt requested on a tower built with mask_prompt_depth=0 (no mask_embedding)")
            x = self.embed(A, M, patch_open)
            for i in range(last):
                x, st = self.run_blocks(x, i, i + 1, stats=st, with_stats=True)
                if i + 1 < self.mask_prompt_depth:
                    ops.mask_prompt_select(x, patch_open, self.w["mask_embedding"][i + 1], 1)
                    st = None                                         # rows rewritten in place: statistics handed over by the GEMM are stale
        # the mask prompt after the last block (depth > layers) only rewrites patch tokens, which nothing reads any more
        return self.head(self.last_block_cls(x, last, st))


class DeviceCrops:
    """Crop bookkeeping of a forward whose crop list was built on the device (MODEL.CLIP_ADAPTER.CROP_LIST): everything stays a device
    tensor of data-independent shape -- crops [T*Q,6], slot [T,Q] (logit row or -1), counts [1] (non-empty masks)."""

    def __init__(self, crops, slot, counts):
        self.crops, self.slot, self.counts = crops, slot, counts

    def valid_host(self):
        """numpy bool [T,Q] (synchronises: tests / stage dumps only)"""
        return (self.slot >= 0).cpu().numpy()


class ClipAdapter:
    mask_prompt_depth = 0            # AdaptedClipAdapter: > 0 (the tower owns a mask_embedding)
    mask_prompt_fwd = False
    crop_list = "auto"               # MODEL.CLIP_ADAPTER.CROP_LIST (config.py)
    # auto mode, PER HOST THREAD (ClipPipeline / --streams run several forwards of one model at once: a shared queue would be an unguarded
    # read-modify-write, and a dropped or duplicated entry makes the host / device choice -- and the GEMM shapes -- timing-dependent again):
    #   _valid_frac     share of non-empty masks of the most recent clip of THIS thread whose count is known
    #   _pending_counts (event, pinned host int32 [1], T*Q) of this thread's device-list forwards whose counts are still on their way
    def _auto(self):
        """this adapter's per-host-thread state (created on first use; dict.setdefault is atomic, so two threads agree on one object)"""
        return self.__dict__.setdefault("_auto_local", __import__("threading").local())

    # a thread that has no share of its own yet (the slot threads of a ClipPipeline.run are new every call) starts from the newest share ANY
    # thread of this adapter has published: one float, written whole -- the queue of pending counts is what must not be shared
    _valid_frac = property(lambda self: getattr(self._auto(), "frac", self.__dict__.get("_last_frac")),
                           lambda self, v: (setattr(self._auto(), "frac", v), self.__dict__.__setitem__("_last_frac", v))[0])
    _pending_counts = property(lambda self: getattr(self._auto(), "pend", ()), lambda self, v: setattr(self._auto(), "pend", v))

    def __init__(self, clip_model_name="ViT-B/16", text_templates="vild", arch=None, precision="fp16"):
        self.clip_model_name = clip_model_name
        self.arch = dict(arch or _CLIP_ARCH[clip_model_name])
        self.precision = precision
        self.visual = ClipVisual(**self.arch, precision=precision, mask_prompt_depth=self.mask_prompt_depth)
        self.input_resolution = self.arch["resolution"]
        self.templates = text_templates
        self.text_cache = {}

    def load_state_dict(self, sd, prefix="clip_adapter.", device="cuda"):
        from .text import ClipText
        self.device = device
        self.visual.load_state_dict(sd, prefix + "clip_model.visual.", device)
        self.text_tower = ClipText.from_state_dict(sd, prefix + "clip_model.", device)     # None: vision-only checkpoint
        self.tokenizer = None
        return self

    # ---- text side -------------------------------------------------------------------------
    def set_text_features(self, noun_list, feats):
        """feats [K, embed_dim] unit-norm rows (what encode_text would cache, adapter.py:121-138)."""
        feats = feats.float().to(self.device)
        self.text_cache.update(dict(zip(noun_list, feats)))

    def set_tokenizer(self, tokenizer):
        """tokenizer: openvis_amd.simple_tokenizer.SimpleTokenizer (needs CLIP's public BPE merge table)."""
        self.tokenizer = tokenizer

    def encode_text(self, noun_list):
        """adapter.py:121-138: prompt-ensemble embeddings of the words not cached yet (text tower on the GPU), then the
        cached unit rows [K, embed_dim]."""
        missing = [w for w in noun_list if w not in self.text_cache]
        if missing:
            if getattr(self, "text_tower", None) is None:
                raise NotImplementedError(
                    f"no cached text embedding for {missing[:3]}... and the checkpoint has no CLIP text tower: "
                    "call ClipAdapter.set_text_features(class_names, embeddings) first")
            from .text import PREDEFINED_TEMPLATES, encode_nouns
            if self.tokenizer is None:
                from ...simple_tokenizer import SimpleTokenizer
                self.tokenizer = SimpleTokenizer()
            templates = PREDEFINED_TEMPLATES[self.templates] if isinstance(self.templates, str) else self.templates
            self.text_cache.update(dict(zip(missing, encode_nouns(self.text_tower, self.tokenizer, templates, missing))))
        return torch.stack([self.text_cache[w] for w in noun_list]).contiguous()

    # ---- image side ------------------------------------------------------------------------
    def preprocess_boxes(self, masks_lowres, Hp, Wp):
        """valid flags + crop list from the mask logits (adapter.py:86-102). One small D2H copy, as the reference's
        BitMasks.get_bounding_boxes host loop (adapter.py:94)."""
        boxes = ops.mask_bbox(masks_lowres, Hp, Wp).cpu().numpy()             # [T,Q,4]
        valid = boxes[..., 2] >= 0                                            # [T,Q]
        tq = np.argwhere(valid)                                               # (t, q) lexicographic
        crops = np.concatenate([tq, boxes[valid]], axis=1).astype(np.int32) if len(tq) else np.zeros((0, 6), np.int32)
        return valid, crops

    def _use_device_list(self):
        """MODEL.CLIP_ADAPTER.CROP_LIST resolved for this forward; "auto" looks at the newest known share of non-empty masks."""
        if self.crop_list != "auto":
            return self.crop_list == "device"
        # Deterministic: clip n looks at the count of clip n - 2, never at "whatever has arrived" (an event.query() made the choice -- and with
        # it the GEMM shapes: M compacted or T Q -- depend on host timing).  That copy was queued two forwards ago behind clip n - 2's mask
        # kernel; the launch window (_lib.LAUNCH_WINDOW) keeps the host less than a clip ahead, so the synchronize normally returns at once
        # (it IS a host wait in the launch path if the host ever runs two clips ahead: bounded by one clip's GPU time).
        if len(self._pending_counts) >= 2:
            ev, host, n = self._pending_counts[0]
            ev.synchronize()
            self._valid_frac = float(host[0]) / float(n)
            self._pending_counts = self._pending_counts[1:]
        return self._valid_frac is not None and self._valid_frac >= 0.9

    def forward(self, frames, text, masks_lowres, padded_hw):
        """frames uint8 [T,3,H,W] (device); masks_lowres [Q,T,h,w] logits; returns (sim_logits [M,K] or None, valid [T,Q],
        crops int32 [M,6]) -- host crop list -- or (sim_logits [T*Q,K], DeviceCrops, DeviceCrops.crops) -- device crop list."""
        Hp, Wp = padded_hw
        if self._use_device_list():
            boxes = ops.mask_bbox(masks_lowres, Hp, Wp)                       # [T,Q,4] stays on the device
            crops_d, slot, counts = ops.crop_list_static(boxes, Hp, Wp)
            dc = DeviceCrops(crops_d, slot, counts)
            if self.crop_list == "auto":                                      # the count rides back on the hand-off stream, nobody waits for it
                from ...output import copy_stream
                side = copy_stream(counts.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                    host.copy_(counts, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                counts.record_stream(side)
                self._pending_counts = tuple(self._pending_counts) + ((ev, host, slot.numel()),)
            M = crops_d.shape[0]
            if self.mask_prompt_fwd:
                A, patch_open = ops.clip_crop_patches_masked(frames, masks_lowres, crops_d, Hp, Wp, self.input_resolution, self.arch["patch"],
                                                             PIXEL_MEAN, PIXEL_STD, out_f16=(self.precision == "fp16"))
                feat = self.visual.forward_patches(A, M, patch_open)
            else:
                A = ops.clip_crop_patches(frames, masks_lowres, crops_d, Hp, Wp, self.input_resolution, self.arch["patch"], PIXEL_MEAN, PIXEL_STD,
                                          out_f16=(self.precision == "fp16"))
                feat = self.visual.forward_patches(A, M)
            feat = ops.l2norm_rows(feat, 100.0)
            return ops.gemm_nt(feat, self.encode_text(text)), dc, crops_d
        valid, crops = self.preprocess_boxes(masks_lowres, Hp, Wp)
        self._valid_frac = float(valid.mean()) if valid.size else 0.0
        self._pending_counts = ()                                 # this clip's own share is newer than any count still on its way
        if crops.shape[0] == 0:
            return None, valid, crops
        crops_d = ops.to_device_async(crops, self.device)
        if self.mask_prompt_fwd:                                              # mask_adapted_adapter.py:68-69
            A, patch_open = ops.clip_crop_patches_masked(frames, masks_lowres, crops_d, Hp, Wp, self.input_resolution,
                                                         self.arch["patch"], PIXEL_MEAN, PIXEL_STD,
                                                         out_f16=(self.precision == "fp16"))
            feat = self.visual.forward_patches(A, crops.shape[0], patch_open)
        else:
            A = ops.clip_crop_patches(frames, masks_lowres, crops_d, Hp, Wp, self.input_resolution, self.arch["patch"],
                                      PIXEL_MEAN, PIXEL_STD, out_f16=(self.precision == "fp16"))
            feat = self.visual.forward_patches(A, crops.shape[0])
        text_features = self.encode_text(text)
        feat = ops.l2norm_rows(feat, 100.0)                                   # normalize, then temperature (:144,146)
        return ops.gemm_nt(feat, text_features), valid, crops

    __call__ = forward


class _NonObjectMixin:
    """Bg* adapters (adapter.py:150-161; mask_adapted_adapter.py:154-166): one learned, L2-normalised "non-object" row is
    appended to the class embeddings, so sim_logits has K + 1 columns."""

    def load_state_dict(self, sd, prefix="clip_adapter.", device="cuda"):
        super().load_state_dict(sd, prefix, device)
        e = sd[prefix + "non_object_embedding"].float().to(device).reshape(1, -1)
        self.non_object_embedding = ops.l2norm_rows(e.contiguous())
        return self

    def encode_text(self, noun_list):
        return torch.cat([super().encode_text(noun_list), self.non_object_embedding], dim=0).contiguous()


class BgClipAdapter(_NonObjectMixin, ClipAdapter):
    pass


class AdaptedClipAdapter(ClipAdapter):
    """mask_adapted_adapter.py:35-148: same crops as ClipAdapter (its _preprocess_image :79-123 is adapter.py:73-116 plus
    the mask regions), CLIP tower built with `mask_prompt_depth` mask embeddings; with `mask_prompt_fwd` the tower gets the
    mask regions (model.py:331-352), otherwise it runs as the plain ViT."""

    def __init__(self, clip_model_name="ViT-B/16", mask_prompt_depth=3, mask_prompt_fwd=True, text_templates="vild", arch=None,
                 precision="fp16"):
        self.mask_prompt_depth = int(mask_prompt_depth)
        self.mask_prompt_fwd = bool(mask_prompt_fwd)
        if self.mask_prompt_fwd and self.mask_prompt_depth < 1:
            # the reference indexes mask_embedding[0] of an empty parameter here (model.py:338) and raises
            raise ValueError("MASK_PROMPT_FWD needs MASK_PROMPT_DEPTH >= 1")
        super().__init__(clip_model_name, text_templates, arch, precision)


class BgAdaptedClipAdapter(_NonObjectMixin, AdaptedClipAdapter):
    pass
