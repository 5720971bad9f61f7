"""VideoMultiScaleMaskedTransformerDecoder — mirror of
openvis/modeling/transformer_decoder/video_mask2former_transformer_decoder.py:218-471 (eval path, post-norm).

Same constructor arguments / state-dict keys.  MI355X restructuring that does not change results beyond f32 rounding:
  * the boolean attention mask is built once per layer as uint8 [Q, keys] and shared by the 8 heads (:468 repeats it);
  * intermediate prediction heads evaluate mask logits only at the attention-target resolution, by pooling the mask
    FEATURES with the exact 2x2-centre-tap mean that F.interpolate(bilinear, 1/s) applies to the logits (:463-466);
    only the final head produces full-resolution masks (aux_outputs are training-only, :445-451);
  * K/V of a level are projected with one GEMM each per layer; scores are never materialised (flash attention)."""
import torch

from ...config import decoder_precision as _decoder_precision
from ... import ops
from ...registry import TRANSFORMER_DECODER_REGISTRY


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    name = cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME
    return TRANSFORMER_DECODER_REGISTRY.get(name).from_config(cfg, in_channels, mask_classification)


@TRANSFORMER_DECODER_REGISTRY.register()
class VideoMultiScaleMaskedTransformerDecoder:
    def __init__(self, in_channels, mask_classification=True, *, num_classes, hidden_dim, num_queries, nheads,
                 dim_feedforward, dec_layers, pre_norm, mask_dim, enforce_input_project, num_frames, precision="fp16"):
        if pre_norm:
            raise NotImplementedError("PRE_NORM=True is not used by any reference config")
        if in_channels != hidden_dim or enforce_input_project:
            raise NotImplementedError("input_proj conv (in_channels != hidden_dim) is not used by any reference config")
        self.mask_classification = mask_classification
        self.num_frames, self.num_heads, self.num_layers = num_frames, nheads, dec_layers
        self.num_queries, self.hidden_dim, self.num_feature_levels = num_queries, hidden_dim, 3
        # "fp16": Linear / einsum operands rounded to fp16 with f32 accumulation (the reference decoder runs under
        # autocast, train_net.py:241); LayerNorm, softmax, residuals f32.  "fp32": exact-f32 MFMA everywhere.
        self.precision = precision
        # (x + query_embed) W^T + b as x W^T + constant (load_state_dict); False: add kernel + separate q | k and v GEMMs.  Off under the autocast
        # policy, where the reference rounds the SUM to fp16 before the Linear
        self.fold_query_pos = precision != "fp16"
        self.fuse_heads = precision != "fp16"   # decoder_norm + mask-embedding MLP as one launch (f32 arithmetic; the autocast policy keeps the GEMMs)
        self.batch_kv = True           # K / V projections of a level's three layers as one GEMM each (load_state_dict); False: one per layer
        self.w = {}
        self.h = {}
        self._pos_cache = {}

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        assert cfg.MODEL.MASK_FORMER.DEC_LAYERS >= 1
        return cls(in_channels, mask_classification, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                   hidden_dim=cfg.MODEL.MASK_FORMER.HIDDEN_DIM, num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
                   nheads=cfg.MODEL.MASK_FORMER.NHEADS, dim_feedforward=cfg.MODEL.MASK_FORMER.DIM_FEEDFORWARD,
                   dec_layers=cfg.MODEL.MASK_FORMER.DEC_LAYERS - 1, pre_norm=cfg.MODEL.MASK_FORMER.PRE_NORM,
                   mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM, enforce_input_project=cfg.MODEL.MASK_FORMER.ENFORCE_INPUT_PROJ,
                   num_frames=cfg.INPUT.SAMPLING_FRAME_NUM,
                   precision=_decoder_precision(cfg))

    def load_state_dict(self, sd, prefix="sem_seg_head.predictor.", device="cuda"):
        sd = dict(sd)
        for k in list(sd.keys()):                                   # :224-245 static_query -> query_feat
            if k.startswith(prefix) and "static_query" in k:
                sd[k.replace("static_query", "query_feat")] = sd.pop(k)
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        self.device = device
        C = self.hidden_dim
        w = self.w
        for i in range(self.num_layers):
            cp = f"transformer_cross_attention_layers.{i}."
            ipw, ipb = g(cp + "multihead_attn.in_proj_weight"), g(cp + "multihead_attn.in_proj_bias")
            w[f"ca{i}.wq"], w[f"ca{i}.bq"] = ipw[:C].contiguous(), ipb[:C].contiguous()
            w[f"ca{i}.wk"], w[f"ca{i}.bk"] = ipw[C:2 * C].contiguous(), ipb[C:2 * C].contiguous()
            w[f"ca{i}.wv"], w[f"ca{i}.bv"] = ipw[2 * C:].contiguous(), ipb[2 * C:].contiguous()
            w[f"ca{i}.wo"], w[f"ca{i}.bo"] = g(cp + "multihead_attn.out_proj.weight"), g(cp + "multihead_attn.out_proj.bias")
            w[f"ca{i}.nw"], w[f"ca{i}.nb"] = g(cp + "norm.weight"), g(cp + "norm.bias")
            sp = f"transformer_self_attention_layers.{i}."
            ipw, ipb = g(sp + "self_attn.in_proj_weight"), g(sp + "self_attn.in_proj_bias")
            w[f"sa{i}.wqk"], w[f"sa{i}.bqk"] = ipw[:2 * C].contiguous(), ipb[:2 * C].contiguous()
            w[f"sa{i}.wv"], w[f"sa{i}.bv"] = ipw[2 * C:].contiguous(), ipb[2 * C:].contiguous()
            w[f"sa{i}.wo"], w[f"sa{i}.bo"] = g(sp + "self_attn.out_proj.weight"), g(sp + "self_attn.out_proj.bias")
            w[f"sa{i}.nw"], w[f"sa{i}.nb"] = g(sp + "norm.weight"), g(sp + "norm.bias")
            fp = f"transformer_ffn_layers.{i}."
            for k in ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm.weight", "norm.bias"):
                w[f"ffn{i}.{k}"] = g(fp + k)
        # K / V projections of the memory, batched per feature level: the layers i, i + L, i + 2L attend to the same level, so their K (V)
        # projections are one GEMM [keys, C] x [C, 3C] -- the level is read once instead of three times and the launch is large enough for the
        # ping-pong kernel (5 x 92 x 160 keys: 2 x 0.11 ms instead of 6 x 0.067).  Same dot products per output element.
        L = self.num_feature_levels
        for li in range(L):
            ids = [i for i in range(self.num_layers) if i % L == li]
            for n in ("k", "v"):
                w[f"ca{n}_lvl{li}.w"] = torch.cat([w[f"ca{i}.w{n}"] for i in ids]).contiguous()
                w[f"ca{n}_lvl{li}.b"] = torch.cat([w[f"ca{i}.b{n}"] for i in ids]).contiguous()
        for k in ("decoder_norm.weight", "decoder_norm.bias", "query_feat.weight", "query_embed.weight", "level_embed.weight"):
            w[k] = g(k)
        # with_pos_embed(tgt, query_pos) in front of a Linear (cross-attention query :110-113, self-attention q = k :52-55) is linear in the
        # CONSTANT query embedding: (x + e) W^T + b = x W^T + (e W^T + b).  The second term is folded here (f64 accumulate, stored f32) and rides
        # as the GEMM's residual: no add kernel, and the self-attention's q | k and v projections become one GEMM [Q, C] x [C, 3C].
        e64 = w["query_embed.weight"].double()
        for i in range(self.num_layers):
            w[f"ca{i}.rq"] = (e64 @ w[f"ca{i}.wq"].double().t() + w[f"ca{i}.bq"].double()).float().contiguous()
            w[f"sa{i}.wqkv"] = torch.cat([w[f"sa{i}.wqk"], w[f"sa{i}.wv"]]).contiguous()
            rqk = e64 @ w[f"sa{i}.wqk"].double().t() + w[f"sa{i}.bqk"].double()
            w[f"sa{i}.rqkv"] = torch.cat([rqk, w[f"sa{i}.bv"].double().expand(rqk.shape[0], -1)], 1).float().contiguous()
        self._rep_cache = {}
        self._load_class_head(g)
        for j in range(3):
            w[f"mask_embed.{j}.w"], w[f"mask_embed.{j}.b"] = g(f"mask_embed.layers.{j}.weight"), g(f"mask_embed.layers.{j}.bias")
            w[f"mask_embed.{j}.wt"] = w[f"mask_embed.{j}.w"].t().contiguous()          # [in, out]: operand layout of ops.ln_mlp3
        self._pos_cache.clear()
        self.h = {k: ops.cast_f16(v) for k, v in w.items() if v.dim() == 2 and v.shape[1] % 8 == 0 and
                  (k.split(".")[-1].startswith("w") or k.endswith("weight")) and "query" not in k and "level_embed" not in k} \
            if self.precision == "fp16" else {}
        return self

    # ---- the class head: nn.Linear(hidden_dim, num_classes + 1) here (video decoder:303-304); the Embedding* / Proposal* variants below
    # replace these two methods only (video decoder:487-537, frame decoder:157-207)
    def _load_class_head(self, g):
        if self.mask_classification:
            self.w["class_embed.weight"], self.w["class_embed.bias"] = g("class_embed.weight"), g("class_embed.bias")

    def _class_head(self, dec):
        return ops.gemm_nt(dec, self.w["class_embed.weight"], self.w["class_embed.bias"])

    def _rep(self, key, T):
        """Constant residual w[key] [Q, N] repeated for the T frames of a per-frame decoder ([T * Q, N]); T = 1: the tensor itself."""
        if T == 1:
            return self.w[key]
        hit = self._rep_cache.get((key, T))
        if hit is None:
            hit = self._rep_cache[(key, T)] = self.w[key].repeat(T, 1).contiguous()
        return hit

    def _mm(self, x, wk, bk=None, residual=None, act=ops.ACT_NONE):
        return ops.gemm_nt(x, self.w[wk], self.w[bk] if bk else None, residual, act, w16=self.h.get(wk), cw=True)

    def _pos(self, T, H, W):
        key = (T, H, W)
        if key not in self._pos_cache:      # PositionEmbeddingSine3D (position_encoding.py:135-165), input independent
            self._pos_cache[key] = ops.pe_sine(T, H, W, self.hidden_dim // 2, True, None, self.device).view(T * H * W, -1)
        return self._pos_cache[key]

    def _mask_embed(self, output):
        w = self.w
        if self.fuse_heads and output.shape[-1] == 256 and all(w[f"mask_embed.{j}.w"].shape == (256, 256) for j in range(3)):
            return ops.ln_mlp3(output, w["decoder_norm.weight"], w["decoder_norm.bias"], [w[f"mask_embed.{j}.wt"] for j in range(3)],
                               [w[f"mask_embed.{j}.b"] for j in range(3)])
        dec = ops.layernorm(output, w["decoder_norm.weight"], w["decoder_norm.bias"])
        h = self._mm(dec, "mask_embed.0.w", "mask_embed.0.b", None, ops.ACT_RELU)
        h = self._mm(h, "mask_embed.1.w", "mask_embed.1.b", None, ops.ACT_RELU)
        return dec, self._mm(h, "mask_embed.2.w", "mask_embed.2.b")

    @staticmethod
    def _nsplit(nk):
        return max(1, min(64, nk // 256))        # x 8 heads: >= 128 workgroups from 4 600 keys on (nk // 1024 left the 18 400-key level on 136)

    def forward(self, x, mask_features, mask=None, shard=None):
        """x: 3 maps [T,H_l,W_l,C] (res5,res4,res3 scale); mask_features [T,h,w,C] (NHWC).
        Returns {'pred_logits': [1,Q,C+1], 'pred_masks': [1,Q,T,h,w]} (eval: bs = 1, t = T; :381-383).

        shard = (T_total, b0, exchange): the clip's frames are split over GPUs (OpenVIS row "split-KV") and x / mask_features
        hold the frames [b0, b0 + T) of T_total.  The queries are replicated; the cross-attention's keys are this GPU's frames, its flash
        partial is exchanged once per layer (`exchange`: packed f32 [n] -> [R, n], an all-gather) and merged (ops.attention_merge); the
        prediction heads' masks are per-frame products of the replicated mask embedding, so pred_masks covers the local frames only."""
        w = self.w
        T, hm, wm, C = mask_features.shape
        Q, H8 = self.num_queries, self.num_heads
        D = C // H8
        src, kin, sizes, pooled = [], [], [], []
        for i in range(self.num_feature_levels):
            _, H, W, _ = x[i].shape
            sizes.append((H, W))
            s = ops.add_bcast(x[i].reshape(T * H * W, C), w["level_embed.weight"][i].contiguous())   # :401
            src.append(s)
            if shard is None:
                pos = self._pos(T, H, W)
            else:                                                   # the 3-D sine embedding normalises z by the clip's length: slice the clip's table
                pos = self._pos(shard[0], H, W)[shard[1] * H * W:(shard[1] + T) * H * W]
            kin.append(ops.add_bcast(s, pos))                                                          # memory + pos
            sc = hm // H
            pooled.append(ops.center_pool(mask_features, sc).view(T * H * W, C) if sc > 1 else mask_features.view(-1, C))
        f16 = self.precision == "fp16"
        pooled16 = [ops.cast_f16(p) for p in pooled] if f16 else [None] * len(pooled)       # einsum operands under autocast
        query_embed = w["query_embed.weight"]
        output = w["query_feat.weight"]

        def head_mask(out, level):
            _, me = self._mask_embed(out)
            logits = ops.gemm_nt(me, pooled[level], w16=pooled16[level])   # [Q, T*H_l*W_l]
            return ops.attn_mask_from_logits(logits)

        amask, row_open = head_mask(output, 0)
        kv_lvl = {}
        for i in range(self.num_layers):
            li = i % self.num_feature_levels
            Nk = src[li].shape[0]
            # masked cross-attention (:417-426, CrossAttentionLayer.forward_post :110-122)
            if self.fold_query_pos:
                qp = self._mm(output, f"ca{i}.wq", None, w[f"ca{i}.rq"])
            else:
                qp = self._mm(ops.add_bcast(output, query_embed), f"ca{i}.wq", f"ca{i}.bq")
            if self.batch_kv:
                if li not in kv_lvl:
                    kv_lvl[li] = (self._mm(kin[li], f"cak_lvl{li}.w", f"cak_lvl{li}.b"), self._mm(src[li], f"cav_lvl{li}.w", f"cav_lvl{li}.b"))
                kp, vp = (t[:, (i // self.num_feature_levels) * C:] for t in kv_lvl[li])       # column block of layer i in [keys, 3C]
                ldkv = kv_lvl[li][0].shape[1]
            else:
                kp = self._mm(kin[li], f"ca{i}.wk", f"ca{i}.bk")
                vp = self._mm(src[li], f"ca{i}.wv", f"ca{i}.bv")
                ldkv = C
            if shard is None:
                att = ops.attention(qp, kp, vp, 1, H8, Q, Nk, D, 0, C, 0, ldkv, 0, ldkv, amask, row_open, self._nsplit(Nk))
            else:
                part = ops.attention_partial(qp, kp, vp, 1, H8, Q, Nk, D, 0, C, 0, ldkv, 0, ldkv, amask, row_open, self._nsplit(Nk))
                att = ops.attention_merge(shard[2](part), 1, H8, Q, D)
            y = self._mm(att.view(Q, C), f"ca{i}.wo", f"ca{i}.bo", output)
            output = ops.layernorm(y, w[f"ca{i}.nw"], w[f"ca{i}.nb"])
            # self-attention (:428-432, SelfAttentionLayer.forward_post :52-62)
            if self.fold_query_pos:
                qkv = self._mm(output, f"sa{i}.wqkv", None, w[f"sa{i}.rqkv"])                 # [Q, 3C] = q | k | v
                att = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], 1, H8, Q, Q, D, 0, 3 * C, 0, 3 * C, 0, 3 * C)
            else:
                qk = self._mm(ops.add_bcast(output, query_embed), f"sa{i}.wqk", f"sa{i}.bqk")   # [Q, 2C] = q | k
                vv = self._mm(output, f"sa{i}.wv", f"sa{i}.bv")
                att = ops.attention(qk, qk[:, C:], vv, 1, H8, Q, Q, D, 0, 2 * C, 0, 2 * C, 0, C)
            y = self._mm(att.view(Q, C), f"sa{i}.wo", f"sa{i}.bo", output)
            output = ops.layernorm(y, w[f"sa{i}.nw"], w[f"sa{i}.nb"])
            # FFN (:434-437, FFNLayer.forward_post :175-179)
            hdn = self._mm(output, f"ffn{i}.linear1.weight", f"ffn{i}.linear1.bias", None, ops.ACT_RELU)
            y = self._mm(hdn, f"ffn{i}.linear2.weight", f"ffn{i}.linear2.bias", output)
            output = ops.layernorm(y, w[f"ffn{i}.norm.weight"], w[f"ffn{i}.norm.bias"])
            if i + 1 < self.num_layers:
                amask, row_open = head_mask(output, (i + 1) % self.num_feature_levels)
        dec, me = self._mask_embed(output)
        mf2 = mask_features.view(-1, C)
        pred_masks = ops.gemm_nt(me, mf2, w16=ops.cast_f16(mf2) if f16 else None).view(1, Q, T, hm, wm)   # einsum (:460)
        out = {"pred_masks": pred_masks, "pred_embeds": dec}
        if self.mask_classification:
            out["pred_logits"] = self._class_head(dec).view(1, Q, -1)
        return out

    __call__ = forward


class _EmbeddingHead:
    """class_embed = MLP(hidden_dim, 2 clip_dims, clip_dims, 2): the query's CLIP-space embedding (video decoder:487-524, frame decoder:157-193;
    configs/openvoc_ytvis_coco/simplebsl*.yaml)."""

    def _load_class_head(self, g):
        if self.mask_classification:
            for j in range(2):
                self.w[f"class_embed.{j}.w"], self.w[f"class_embed.{j}.b"] = g(f"class_embed.layers.{j}.weight"), g(f"class_embed.layers.{j}.bias")

    def _class_head(self, dec):
        h = ops.gemm_nt(dec, self.w["class_embed.0.w"], self.w["class_embed.0.b"], None, ops.ACT_RELU, w16=self.h.get("class_embed.0.w"), cw=True)
        return ops.gemm_nt(h, self.w["class_embed.1.w"], self.w["class_embed.1.b"], w16=self.h.get("class_embed.1.w"), cw=True)


class _ProposalHead:
    """class_embed = nn.Linear(hidden_dim, 1 + 1): object / no-object (video decoder:527-537, frame decoder:196-207)."""

    def _load_class_head(self, g):
        if self.mask_classification:
            self.w["class_embed.weight"], self.w["class_embed.bias"] = g("class_embed.weight"), g("class_embed.bias")


def _variant_init(self, base, mask_classification, kwargs):
    base.__init__(self, kwargs.pop("in_channels"), False, **kwargs)       # the reference builds the parent without a class head ...
    self.mask_classification = mask_classification                        # ... and adds its own


@TRANSFORMER_DECODER_REGISTRY.register()
class EmbeddingVideoMultiScaleMaskedTransformerDecoder(_EmbeddingHead, VideoMultiScaleMaskedTransformerDecoder):
    def __init__(self, clip_dims, mask_classification, **kwargs):
        _variant_init(self, VideoMultiScaleMaskedTransformerDecoder, mask_classification, kwargs)
        self.clip_dims = clip_dims

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        base = VideoMultiScaleMaskedTransformerDecoder.from_config.__func__(_Kw, cfg, in_channels, mask_classification)
        return cls(cfg.MODEL.CLIP_ADAPTER.CLIP_EMBED_DIMS, mask_classification, **base)


@TRANSFORMER_DECODER_REGISTRY.register()
class ProposalVideoMultiScaleMaskedTransformerDecoder(_ProposalHead, VideoMultiScaleMaskedTransformerDecoder):
    def __init__(self, mask_classification, **kwargs):
        _variant_init(self, VideoMultiScaleMaskedTransformerDecoder, mask_classification, kwargs)

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        return cls(mask_classification, **VideoMultiScaleMaskedTransformerDecoder.from_config.__func__(_Kw, cfg, in_channels, mask_classification))


class _Kw:
    """from_config of the base class called with this in place of `cls`: returns the constructor's keyword arguments instead of an instance."""

    def __new__(cls, in_channels, mask_classification=True, **kw):
        return dict(in_channels=in_channels, **kw)

