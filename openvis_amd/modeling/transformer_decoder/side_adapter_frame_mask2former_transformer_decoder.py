"""SideAdapterFrameMultiScaleMaskedTransformerDecoder — mirror of
openvis/modeling/transformer_decoder/side_adapter_frame_mask2former_transformer_decoder.py:31-176 (eval path).

The frame decoder plus the SAN attention-bias head: attn_features = attn_mlp(bilinear 1/4 of the mask features)
([T,h/4,w/4, 256*clip_heads], NHWC), attn_embed MLP, class_attn_biases = einsum("bqc,bnchw->bnqhw").  Only the last
prediction head's biases are consumed at eval (aux_outputs are training-only), so they are evaluated once."""
import torch

from ...config import decoder_precision as _decoder_precision
from ... import ops
from ...registry import TRANSFORMER_DECODER_REGISTRY
from .frame_mask2former_transformer_decoder import FrameMultiScaleMaskedTransformerDecoder
from .video_mask2former_transformer_decoder import VideoMultiScaleMaskedTransformerDecoder


class SideAdapterHeadMixin:
    """attn_mlp / attn_embed weights and the class-attention-bias head shared by the frame and the video variant."""

    def _load_side_head(self, sd, prefix, device):
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        extra = {}
        for j in range(3):
            extra[f"attn_embed.{j}.w"], extra[f"attn_embed.{j}.b"] = g(f"attn_embed.layers.{j}.weight"), g(f"attn_embed.layers.{j}.bias")
            wt = g(f"attn_mlp.layers.{j}.weight")
            extra[f"attn_mlp.{j}.w"], extra[f"attn_mlp.{j}.b"] = wt.view(wt.shape[0], wt.shape[1]).contiguous(), g(f"attn_mlp.layers.{j}.bias")
        return extra

    def _side_bias_head(self, dec, mask_features, per_frame_queries):
        """dec: decoder_norm(output) rows -- [T*Q, C] (frame decoder: one query set per frame) or [Q, C] (video decoder: one
        query set for the clip).  Returns (class_attn_biases [1,T,n,Q,ha,wa], attn_feats [T,ha,wa,n*C])."""
        T, hm, wm, C = mask_features.shape
        Q, n = self.num_queries, self.clip_heads
        # attn_features: bilinear x0.25 (== 2x2 centre-tap mean of every 4x4 cell) + 3 x conv1x1 (side-frame:67-71)
        af = ops.center_pool(mask_features, 4)
        ha, wa = af.shape[1], af.shape[2]
        af = self._mm(af.view(-1, C), "attn_mlp.0.w", "attn_mlp.0.b", None, ops.ACT_RELU)
        af = self._mm(af, "attn_mlp.1.w", "attn_mlp.1.b", None, ops.ACT_RELU)
        af = self._mm(af, "attn_mlp.2.w", "attn_mlp.2.b")                     # [T*ha*wa, n*C], channel = head*C + c
        ae = self._mm(dec, "attn_embed.0.w", "attn_embed.0.b", None, ops.ACT_RELU)
        ae = self._mm(ae, "attn_embed.1.w", "attn_embed.1.b", None, ops.ACT_RELU)
        ae = self._mm(ae, "attn_embed.2.w", "attn_embed.2.b").view(-1, Q, C)
        # einsum("bqc,bnchw->bnqhw"): per frame, n independent GEMMs [Q,C] x [ha*wa, C]^T (B rows strided by n*C)
        biases = torch.empty((T, n, Q, ha, wa), dtype=torch.float32, device=af.device)
        npix = ha * wa
        af3 = af.view(T, npix, n * C)
        af16 = ops.cast_f16(af3) if self.precision == "fp16" else None
        for t in range(T):
            ops.gemm_nt_batched(ae[t if per_frame_queries else 0], af3[t], biases[t], n, Q, npix, C, C, 0, n * C, C, npix, Q * npix,
                                b16=af16[t] if af16 is not None else None)
        return biases.unsqueeze(0), af3.view(T, ha, wa, n * C)


def _side_from_config(cls, cfg, in_channels, mask_classification):
    assert cfg.MODEL.MASK_FORMER.DEC_LAYERS >= 1
    return cls(cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS, mask_classification, in_channels=in_channels,
               num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, hidden_dim=cfg.MODEL.MASK_FORMER.HIDDEN_DIM,
               num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES, nheads=cfg.MODEL.MASK_FORMER.NHEADS,
               dim_feedforward=cfg.MODEL.MASK_FORMER.DIM_FEEDFORWARD, dec_layers=cfg.MODEL.MASK_FORMER.DEC_LAYERS - 1,
               pre_norm=cfg.MODEL.MASK_FORMER.PRE_NORM, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
               enforce_input_project=cfg.MODEL.MASK_FORMER.ENFORCE_INPUT_PROJ, num_frames=cfg.INPUT.SAMPLING_FRAME_NUM,
               precision=_decoder_precision(cfg))


@TRANSFORMER_DECODER_REGISTRY.register()
class SideAdapterFrameMultiScaleMaskedTransformerDecoder(SideAdapterHeadMixin, FrameMultiScaleMaskedTransformerDecoder):
    def __init__(self, clip_heads, mask_classification, **kwargs):
        super().__init__(mask_classification=False, **kwargs)      # no class_embed (side-frame decoder:35)
        self.clip_heads = clip_heads

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        return _side_from_config(cls, cfg, in_channels, mask_classification)

    def load_state_dict(self, sd, prefix="sem_seg_head.predictor.", device="cuda"):
        extra = self._load_side_head(sd, prefix, device)
        super().load_state_dict(sd, prefix, device)
        self.w.update(extra)
        if self.precision == "fp16":
            self.h.update({k: ops.cast_f16(v) for k, v in extra.items() if k.endswith(".w")})
        return self

    def forward(self, x, mask_features, mask=None):
        out = super().forward(x, mask_features, mask)
        T, C = mask_features.shape[0], mask_features.shape[-1]
        out["class_attn_biases"], out["attn_feats"] = self._side_bias_head(out["pred_embeds"].view(T * self.num_queries, C),
                                                                         mask_features, True)
        return out

    __call__ = forward


@TRANSFORMER_DECODER_REGISTRY.register()
class SideAdapterVideoMultiScaleMaskedTransformerDecoder(SideAdapterHeadMixin, VideoMultiScaleMaskedTransformerDecoder):
    """openvis/modeling/transformer_decoder/side_adapter_video_mask2former_transformer_decoder.py:28-142 (eval path): the
    offline (clip-level) decoder with the SAN bias head -- einsum("bqc,btnchw->btnqhw"), one query set for all frames."""

    def __init__(self, clip_heads, mask_classification, **kwargs):
        super().__init__(mask_classification=False, **kwargs)
        self.clip_heads = clip_heads

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        return _side_from_config(cls, cfg, in_channels, mask_classification)

    def load_state_dict(self, sd, prefix="sem_seg_head.predictor.", device="cuda"):
        extra = self._load_side_head(sd, prefix, device)
        super().load_state_dict(sd, prefix, device)
        self.w.update(extra)
        if self.precision == "fp16":
            self.h.update({k: ops.cast_f16(v) for k, v in extra.items() if k.endswith(".w")})
        return self

    def forward(self, x, mask_features, mask=None):
        out = super().forward(x, mask_features, mask)
        out["class_attn_biases"], out["attn_feats"] = self._side_bias_head(out["pred_embeds"], mask_features, False)
        return out

    __call__ = forward
