"""MSDeformAttnPixelDecoder — mirror of openvis/modeling/pixel_decoder/msdeformattn.py:179-380 (eval path).

forward_features(features) -> (mask_features, out[0], multi_scale_features) like the reference, but every tensor
is channel-last (NHWC / [T, tokens, C]) and all arithmetic runs on the gfx950 kernels.  State-dict keys are the
reference's."""
import numpy as np
import torch

from ... import ops
from ...registry import SEM_SEG_HEADS_REGISTRY
from .ops.modules import MSDeformAttn


def build_pixel_decoder(cfg, input_shape):
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    model = SEM_SEG_HEADS_REGISTRY.get(name).from_config(cfg, input_shape)
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError("Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
                         f"Please implement forward_features for {name} to only return mask features.")
    return model


class MSDeformAttnTransformerEncoderLayer:
    """msdeformattn.py:107-146."""

    def __init__(self, d_model=256, d_ffn=1024, n_levels=3, n_heads=8, n_points=4):
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.w = {}

    def load_state_dict(self, sd, prefix, device):
        self.self_attn.load_state_dict(sd, prefix + "self_attn.", device)
        for k in ("norm1.weight", "norm1.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                  "norm2.weight", "norm2.bias"):
            self.w[k] = sd[prefix + k].float().contiguous().to(device)

    def forward(self, src, pos, spatial_shapes, level_start_index, shapes_host=None):
        w = self.w
        # with_pos_embed (:138) happens inside: query = src + pos, as an add or -- fp16x2 -- as the row-periodic term of the two-output GEMM.
        # output_proj + residual + norm1, linear2 + residual + norm2: the LayerNorm rides in the GEMM's epilogue where the kernel allows it
        src = self.self_attn.forward_encoder_fused(None, src, spatial_shapes, level_start_index, residual=src, shapes_host=shapes_host,
                                                   norm=(w["norm1.weight"], w["norm1.bias"]), pos=pos)    # :139-141
        h = ops.gemm_nt(src, w["linear1.weight"], w["linear1.bias"], None, ops.ACT_RELU, cw=True)
        return ops.gemm_nt_layernorm(h, w["linear2.weight"], w["linear2.bias"], src, w["norm2.weight"], w["norm2.bias"])   # :118-121, 144-146


@SEM_SEG_HEADS_REGISTRY.register()
class MSDeformAttnPixelDecoder:
    def __init__(self, input_shape, *, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
                 transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                 transformer_in_features=("res3", "res4", "res5"), common_stride=4):
        shapes = sorted(input_shape.items(), key=lambda x: x[1]["stride"])
        self.in_features = [k for k, _ in shapes]
        self.transformer_in_features = [k for k, _ in shapes if k in transformer_in_features]
        self.conv_dim, self.mask_dim, self.common_stride = conv_dim, mask_dim, common_stride
        self.nheads = transformer_nheads
        self.layers = [MSDeformAttnTransformerEncoderLayer(conv_dim, transformer_dim_feedforward,
                                                           len(self.transformer_in_features), transformer_nheads, 4)
                       for _ in range(transformer_enc_layers)]
        stride = min(input_shape[k]["stride"] for k in self.transformer_in_features)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(common_stride))
        self.maskformer_num_feature_levels = 3
        self.w = {}
        self._pos_cache = {}

    @classmethod
    def from_config(cls, cfg, input_shape):
        return cls({k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
                   transformer_dropout=cfg.MODEL.MASK_FORMER.DROPOUT, transformer_nheads=cfg.MODEL.MASK_FORMER.NHEADS,
                   transformer_dim_feedforward=1024, transformer_enc_layers=cfg.MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS,
                   conv_dim=cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
                   norm=cfg.MODEL.SEM_SEG_HEAD.NORM,
                   transformer_in_features=cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
                   common_stride=cfg.MODEL.SEM_SEG_HEAD.COMMON_STRIDE)

    def load_state_dict(self, sd, prefix="sem_seg_head.pixel_decoder.", device="cuda"):
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        self.device = device
        for i in range(len(self.transformer_in_features)):
            w = g(f"input_proj.{i}.0.weight")
            self.w[f"input_proj.{i}.w"] = w.view(w.shape[0], w.shape[1]).contiguous()
            self.w[f"input_proj.{i}.b"] = g(f"input_proj.{i}.0.bias")
            self.w[f"input_proj.{i}.gn_w"], self.w[f"input_proj.{i}.gn_b"] = g(f"input_proj.{i}.1.weight"), g(f"input_proj.{i}.1.bias")
        self.w["level_embed"] = g("transformer.level_embed")
        for i, layer in enumerate(self.layers):
            layer.load_state_dict(sd, f"{prefix}transformer.encoder.layers.{i}.", device)
        w = g("mask_features.weight")
        self.w["mask_features.w"], self.w["mask_features.b"] = w.view(w.shape[0], w.shape[1]).contiguous(), g("mask_features.bias")
        for idx in range(1, self.num_fpn_levels + 1):
            w = g(f"adapter_{idx}.weight")
            self.w[f"adapter_{idx}.w"] = w.view(w.shape[0], w.shape[1]).contiguous()
            self.w[f"adapter_{idx}.gn_w"], self.w[f"adapter_{idx}.gn_b"] = g(f"adapter_{idx}.norm.weight"), g(f"adapter_{idx}.norm.bias")
            self.w[f"layer_{idx}.w"] = g(f"layer_{idx}.weight").permute(0, 2, 3, 1).contiguous()
            self.w[f"layer_{idx}.gn_w"], self.w[f"layer_{idx}.gn_b"] = g(f"layer_{idx}.norm.weight"), g(f"layer_{idx}.norm.bias")
        self._pos_cache.clear()
        return self

    def _pos(self, shapes_list):
        """lvl_pos_embed_flatten [S,C] = sine PE + level_embed, identical for every frame (msdeformattn.py:86-91)."""
        key = tuple(shapes_list)
        if key not in self._pos_cache:
            parts = [ops.pe_sine(1, h, w, self.conv_dim // 2, False, self.w["level_embed"][l].contiguous(), self.device)
                     .view(h * w, self.conv_dim) for l, (h, w) in enumerate(shapes_list)]
            shapes = torch.as_tensor(shapes_list, dtype=torch.long, device=self.device)
            lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
            self._pos_cache[key] = (torch.cat(parts, 0).contiguous(), shapes, lsi)
        return self._pos_cache[key]

    def forward_features(self, features, extra_features=None):
        """features: {res2..res5} NHWC f32.  Returns (mask_features [T,h,w,C], out[0], multi_scale_features[3])."""
        w = self.w
        srcs, shapes_list = [], []
        for idx, f in enumerate(self.transformer_in_features[::-1]):                   # res5, res4, res3
            x = features[f]
            T, H, W, C = x.shape
            y = ops.gemm_nt(x.view(-1, C), w[f"input_proj.{idx}.w"], w[f"input_proj.{idx}.b"], cw=True).view(T, H, W, -1)
            y = ops.groupnorm_nhwc(y, w[f"input_proj.{idx}.gn_w"], w[f"input_proj.{idx}.gn_b"])
            if extra_features is not None:                                             # SAN injection (:338-344)
                ex = extra_features[idx]
                if ex.shape[1:3] == (H, W):
                    y = ops.add_bcast(y, ex.contiguous())
                else:
                    ops.bilinear_resize_add(y, ex.contiguous())
            srcs.append(y.view(T, H * W, -1))
            shapes_list.append((H, W))
        pos, shapes, lsi = self._pos(shapes_list)
        src = torch.cat(srcs, 1).contiguous()                                          # [T,S,C] (copy only)
        for layer in self.layers:
            src = layer.forward(src, pos, shapes, lsi, shapes_host=shapes_list)
        T = src.shape[0]
        outs, start = [], 0
        for (H, W) in shapes_list:
            outs.append(src[:, start:start + H * W].contiguous().view(T, H, W, -1))    # split (copy only)
            start += H * W
        for idx, f in enumerate(self.in_features[: self.num_fpn_levels][::-1]):        # res2
            k = self.num_fpn_levels - idx
            x = features[f]
            T, H, W, C = x.shape
            cur = ops.gemm_nt(x.view(-1, C), w[f"adapter_{k}.w"], cw=True).view(T, H, W, -1)
            # lateral + upsampled top-down map (:369-371), then the 3x3 output convolution (:372): where the ping-pong kernel takes the
            # convolution, the GroupNorm writes its result zero-padded and the convolution walks it as a dense GEMM
            lw = w[f"layer_{k}.w"]
            padded = ops.conv3x3_padded_eligible(T, H, W, cur.shape[-1], lw.shape[0])
            y = ops.groupnorm_nhwc(cur, w[f"adapter_{k}.gn_w"], w[f"adapter_{k}.gn_b"], up_add=outs[-1], pad=padded)
            y = ops.conv3x3_padded(y, lw) if padded else ops.conv2d_nhwc(y, lw, 1, 1, cw=True)
            y = ops.groupnorm_nhwc(y, w[f"layer_{k}.gn_w"], w[f"layer_{k}.gn_b"], relu=True)
            outs.append(y)
        top = outs[-1]
        T, H, W, C = top.shape
        mask_features = ops.gemm_nt(top.view(-1, C), w["mask_features.w"], w["mask_features.b"], cw=True).view(T, H, W, -1)
        return mask_features, outs[0], outs[: self.maskformer_num_feature_levels]
