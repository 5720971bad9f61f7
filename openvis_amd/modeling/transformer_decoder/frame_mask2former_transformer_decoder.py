"""FrameMultiScaleMaskedTransformerDecoder — mirror of
openvis/modeling/transformer_decoder/frame_mask2former_transformer_decoder.py:13-137 (eval path).

Same weights / state-dict keys as the video decoder; every frame is decoded independently (batch = T): 2-D sine
position encoding, per-frame masked cross-attention (one uint8 mask per frame, shared by the 8 heads), per-frame mask
einsum as a batched GEMM.  Also returns pred_embeds = decoder_norm(output) [1,T,Q,C] for the tracker (:123-124)."""
import torch

from ... import ops
from ...registry import TRANSFORMER_DECODER_REGISTRY
from .video_mask2former_transformer_decoder import VideoMultiScaleMaskedTransformerDecoder, _EmbeddingHead, _Kw, _ProposalHead, _variant_init


@TRANSFORMER_DECODER_REGISTRY.register()
class FrameMultiScaleMaskedTransformerDecoder(VideoMultiScaleMaskedTransformerDecoder):

    def _pos2d(self, H, W):
        key = ("2d", H, W)
        if key not in self._pos_cache:      # PositionEmbeddingSine2D (transformer_decoder/position_encoding.py:75-100)
            self._pos_cache[key] = ops.pe_sine(1, H, W, self.hidden_dim // 2, False, None, self.device).view(H * W, -1)
        return self._pos_cache[key]

    def _frame_logits(self, me, feats, feats16, T, Q, n_pix, C, out=None, ldc=None, c_bs=None):
        """einsum("bqc,bchw->bqhw") as T independent GEMMs in one launch."""
        if out is None:
            out = torch.empty((T * Q, n_pix), dtype=torch.float32, device=me.device)
            ldc, c_bs = n_pix, Q * n_pix
        ops.gemm_nt_batched(me, feats, out, T, Q, n_pix, C, C, Q * C, C, n_pix * C, ldc, c_bs, b16=feats16)
        return out

    def forward(self, x, mask_features, mask=None):
        w = self.w
        T, hm, wm, C = mask_features.shape
        Q, H8 = self.num_queries, self.num_heads
        D = C // H8
        f16 = self.precision == "fp16"
        src, kin, sizes, pooled, pooled16 = [], [], [], [], []
        for i in range(self.num_feature_levels):
            _, H, W, _ = x[i].shape
            sizes.append((H, W))
            s = ops.add_bcast(x[i].reshape(T * H * W, C), w["level_embed.weight"][i].contiguous())   # frame:64
            src.append(s)
            kin.append(ops.add_bcast(s, self._pos2d(H, W)))                                            # memory + pos
            sc = hm // H
            p = ops.center_pool(mask_features, sc).view(T * H * W, C) if sc > 1 else mask_features.view(-1, C)
            pooled.append(p)
            pooled16.append(ops.cast_f16(p) if f16 else None)
        query_embed = w["query_embed.weight"]
        output = w["query_feat.weight"].unsqueeze(0).repeat(T, 1, 1).contiguous()      # [T,Q,C] (copy only)

        def head_mask(out, level):
            _, me = self._mask_embed(out)                                                # [T,Q,C]
            n_pix = sizes[level][0] * sizes[level][1]
            logits = self._frame_logits(me, pooled[level], pooled16[level], T, Q, n_pix, C)
            return ops.attn_mask_from_logits(logits)                                     # mask [T*Q, ld], row_open [T*Q]

        amask, row_open = head_mask(output, 0)
        kv_lvl = {}
        for i in range(self.num_layers):
            li = i % self.num_feature_levels
            Nk = sizes[li][0] * sizes[li][1]
            if self.fold_query_pos:                                                       # the constant query-embedding terms as residuals
                qp = self._mm(output.view(T * Q, C), f"ca{i}.wq", None, self._rep(f"ca{i}.rq", T)).view(T, Q, C)
            else:
                qp = self._mm(ops.add_bcast(output, query_embed), f"ca{i}.wq", f"ca{i}.bq")
            if self.batch_kv:                                                            # one K and one V GEMM per level (video decoder, load_state_dict)
                if li not in kv_lvl:
                    kv_lvl[li] = (self._mm(kin[li], f"cak_lvl{li}.w", f"cak_lvl{li}.b"), self._mm(src[li], f"cav_lvl{li}.w", f"cav_lvl{li}.b"))
                kp, vp = (t[:, (i // self.num_feature_levels) * C:] for t in kv_lvl[li])
                ldkv = kv_lvl[li][0].shape[1]
            else:
                kp = self._mm(kin[li], f"ca{i}.wk", f"ca{i}.bk")
                vp = self._mm(src[li], f"ca{i}.wv", f"ca{i}.bv")
                ldkv = C
            # split the keys over workgroups until ~1000 are in flight: with one split a 5-frame clip runs the 14 720-key level on
            # 40 workgroups (231 us per launch on average, 13 % of the SANOnline step)
            nsplit = max(1, min(64, Nk // 256, max(1, 2048 // (T * H8))))
            att = ops.attention(qp, kp, vp, T, H8, Q, Nk, D, Q * C, C, Nk * ldkv, ldkv, Nk * ldkv, ldkv, amask, row_open, nsplit,
                                mask_per_batch=True)
            y = self._mm(att.view(T * Q, C), f"ca{i}.wo", f"ca{i}.bo", output.view(T * Q, C))
            output = ops.layernorm(y, w[f"ca{i}.nw"], w[f"ca{i}.nb"]).view(T, Q, C)
            if self.fold_query_pos:
                qkv = self._mm(output.view(T * Q, C), f"sa{i}.wqkv", None, self._rep(f"sa{i}.rqkv", T))      # [T Q, 3C] = q | k | v
                att = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], T, H8, Q, Q, D, Q * 3 * C, 3 * C, Q * 3 * C, 3 * C, Q * 3 * C, 3 * C)
            else:
                qk = self._mm(ops.add_bcast(output, query_embed), f"sa{i}.wqk", f"sa{i}.bqk").view(T * Q, 2 * C)
                vv = self._mm(output, f"sa{i}.wv", f"sa{i}.bv")
                att = ops.attention(qk, qk[:, C:], vv, T, H8, Q, Q, D, Q * 2 * C, 2 * C, Q * 2 * C, 2 * C, Q * C, C)
            y = self._mm(att.view(T * Q, C), f"sa{i}.wo", f"sa{i}.bo", output.view(T * Q, C))
            output = ops.layernorm(y, w[f"sa{i}.nw"], w[f"sa{i}.nb"]).view(T, Q, C)
            hdn = self._mm(output, f"ffn{i}.linear1.weight", f"ffn{i}.linear1.bias", None, ops.ACT_RELU)
            y = self._mm(hdn, f"ffn{i}.linear2.weight", f"ffn{i}.linear2.bias", output)
            output = ops.layernorm(y, w[f"ffn{i}.norm.weight"], w[f"ffn{i}.norm.bias"]).view(T, Q, C)
            if i + 1 < self.num_layers:
                amask, row_open = head_mask(output, (i + 1) % self.num_feature_levels)
        dec, me = self._mask_embed(output)                                               # [T,Q,C]
        n_pix = hm * wm
        mf2 = mask_features.view(-1, C)
        pred_masks = torch.empty((Q, T, hm, wm), dtype=torch.float32, device=me.device)
        # "(b t) q h w -> b q t h w" (frame:116-117) written directly: row q of frame t starts at (q*T + t)*h*w
        self._frame_logits(me, mf2, ops.cast_f16(mf2) if f16 else None, T, Q, n_pix, C, out=pred_masks, ldc=T * n_pix,
                           c_bs=n_pix)
        out = {"pred_masks": pred_masks.view(1, Q, T, hm, wm), "pred_embeds": dec.view(1, T, Q, C),
               "mask_feats": mask_features, "size_list": sizes}
        if self.mask_classification:
            out["pred_logits"] = self._class_head(dec.view(T * Q, C)).view(1, T, Q, -1)
        return out

    __call__ = forward


@TRANSFORMER_DECODER_REGISTRY.register()
class EmbeddingFrameMultiScaleMaskedTransformerDecoder(_EmbeddingHead, FrameMultiScaleMaskedTransformerDecoder):
    """frame decoder:157-193 (configs/openvoc_ytvis_coco/simplebsl_online*.yaml)."""

    def __init__(self, clip_dims, mask_classification, **kwargs):
        _variant_init(self, FrameMultiScaleMaskedTransformerDecoder, mask_classification, kwargs)
        self.clip_dims = clip_dims

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        base = VideoMultiScaleMaskedTransformerDecoder.from_config.__func__(_Kw, cfg, in_channels, mask_classification)
        return cls(cfg.MODEL.CLIP_ADAPTER.CLIP_EMBED_DIMS, mask_classification, **base)


@TRANSFORMER_DECODER_REGISTRY.register()
class ProposalFrameMultiScaleMaskedTransformerDecoder(_ProposalHead, FrameMultiScaleMaskedTransformerDecoder):
    """frame decoder:196-207."""

    def __init__(self, mask_classification, **kwargs):
        _variant_init(self, FrameMultiScaleMaskedTransformerDecoder, mask_classification, kwargs)

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        return cls(mask_classification, **VideoMultiScaleMaskedTransformerDecoder.from_config.__func__(_Kw, cfg, in_channels, mask_classification))

