"""State-dict shape specs (the reference's key names) and a seeded random initialiser.
There is no network for checkpoints: benchmarks and tests use random-init weights of the named architecture;
a real checkpoint's state dict loads through the same `load_state_dict` paths."""
import math

import torch

from .modeling.clip_adapter.adapter import _CLIP_ARCH


def resnet50_spec(prefix="backbone."):
    spec = []

    def conv(p, cout, cin, k):
        spec.append((p + ".weight", (cout, cin, k, k)))
        for n in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{p}.norm.{n}", (cout,)))

    conv(prefix + "stem.conv1", 64, 3, 7)
    cin = 64
    for name, nblocks, mid, cout in (("res2", 3, 64, 256), ("res3", 4, 128, 512), ("res4", 6, 256, 1024), ("res5", 3, 512, 2048)):
        for i in range(nblocks):
            p = f"{prefix}{name}.{i}"
            if i == 0:
                conv(p + ".shortcut", cout, cin, 1)
            conv(p + ".conv1", mid, cin, 1)
            conv(p + ".conv2", mid, mid, 3)
            conv(p + ".conv3", cout, mid, 1)
            cin = cout
    return spec


def swin_spec(prefix="backbone.", embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=12):
    """backbone/swin.py state dict (float tensors; the integer relative_position_index buffers are derived)."""
    s = [(prefix + "patch_embed.proj.weight", (embed_dim, 3, 4, 4)), (prefix + "patch_embed.proj.bias", (embed_dim,)),
         (prefix + "patch_embed.norm.weight", (embed_dim,)), (prefix + "patch_embed.norm.bias", (embed_dim,))]
    for i, d in enumerate(depths):
        C = embed_dim * 2 ** i
        for j in range(d):
            p = f"{prefix}layers.{i}.blocks.{j}."
            s += [(p + "norm1.weight", (C,)), (p + "norm1.bias", (C,)),
                  (p + "attn.relative_position_bias_table", ((2 * window - 1) ** 2, num_heads[i])),
                  (p + "attn.qkv.weight", (3 * C, C)), (p + "attn.qkv.bias", (3 * C,)), (p + "attn.proj.weight", (C, C)),
                  (p + "attn.proj.bias", (C,)), (p + "norm2.weight", (C,)), (p + "norm2.bias", (C,)),
                  (p + "mlp.fc1.weight", (4 * C, C)), (p + "mlp.fc1.bias", (4 * C,)), (p + "mlp.fc2.weight", (C, 4 * C)),
                  (p + "mlp.fc2.bias", (C,))]
        if i < len(depths) - 1:
            s += [(f"{prefix}layers.{i}.downsample.reduction.weight", (2 * C, 4 * C)),
                  (f"{prefix}layers.{i}.downsample.norm.weight", (4 * C,)), (f"{prefix}layers.{i}.downsample.norm.bias", (4 * C,))]
        s += [(f"{prefix}norm{i}.weight", (C,)), (f"{prefix}norm{i}.bias", (C,))]
    return s


SWIN_ARCH = {"swin_l": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window=12),     # swin/openvis_swinL_*.yaml:5-9
             "swin_b": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window=12)}      # swin/brivis_SwinB_*.yaml:5-9


def pixel_decoder_spec(prefix="sem_seg_head.pixel_decoder.", C=256, layers=6, ffn=1024, M=8, L=3, P=4,
                       in_channels=(256, 512, 1024, 2048)):
    s = []
    for i, cin in enumerate(in_channels[:0:-1]):
        s += [(f"{prefix}input_proj.{i}.0.weight", (C, cin, 1, 1)), (f"{prefix}input_proj.{i}.0.bias", (C,)),
              (f"{prefix}input_proj.{i}.1.weight", (C,)), (f"{prefix}input_proj.{i}.1.bias", (C,))]
    s.append((prefix + "transformer.level_embed", (L, C)))
    for i in range(layers):
        p = f"{prefix}transformer.encoder.layers.{i}."
        s += [(p + "self_attn.sampling_offsets.weight", (M * L * P * 2, C)), (p + "self_attn.sampling_offsets.bias", (M * L * P * 2,)),
              (p + "self_attn.attention_weights.weight", (M * L * P, C)), (p + "self_attn.attention_weights.bias", (M * L * P,)),
              (p + "self_attn.value_proj.weight", (C, C)), (p + "self_attn.value_proj.bias", (C,)),
              (p + "self_attn.output_proj.weight", (C, C)), (p + "self_attn.output_proj.bias", (C,)),
              (p + "norm1.weight", (C,)), (p + "norm1.bias", (C,)),
              (p + "linear1.weight", (ffn, C)), (p + "linear1.bias", (ffn,)),
              (p + "linear2.weight", (C, ffn)), (p + "linear2.bias", (C,)),
              (p + "norm2.weight", (C,)), (p + "norm2.bias", (C,))]
    s += [(prefix + "mask_features.weight", (C, C, 1, 1)), (prefix + "mask_features.bias", (C,)),
          (prefix + "adapter_1.weight", (C, in_channels[0], 1, 1)), (prefix + "adapter_1.norm.weight", (C,)), (prefix + "adapter_1.norm.bias", (C,)),
          (prefix + "layer_1.weight", (C, C, 3, 3)), (prefix + "layer_1.norm.weight", (C,)), (prefix + "layer_1.norm.bias", (C,))]
    return s


def video_decoder_spec(prefix="sem_seg_head.predictor.", C=256, layers=9, ffn=2048, Q=100, num_classes=1):
    s = []
    for i in range(layers):
        for kind, attn in (("transformer_cross_attention_layers", "multihead_attn"), ("transformer_self_attention_layers", "self_attn")):
            p = f"{prefix}{kind}.{i}."
            s += [(p + attn + ".in_proj_weight", (3 * C, C)), (p + attn + ".in_proj_bias", (3 * C,)),
                  (p + attn + ".out_proj.weight", (C, C)), (p + attn + ".out_proj.bias", (C,)),
                  (p + "norm.weight", (C,)), (p + "norm.bias", (C,))]
        p = f"{prefix}transformer_ffn_layers.{i}."
        s += [(p + "linear1.weight", (ffn, C)), (p + "linear1.bias", (ffn,)), (p + "linear2.weight", (C, ffn)),
              (p + "linear2.bias", (C,)), (p + "norm.weight", (C,)), (p + "norm.bias", (C,))]
    s += [(prefix + "decoder_norm.weight", (C,)), (prefix + "decoder_norm.bias", (C,)),
          (prefix + "query_feat.weight", (Q, C)), (prefix + "query_embed.weight", (Q, C)), (prefix + "level_embed.weight", (3, C)),
          (prefix + "class_embed.weight", (num_classes + 1, C)), (prefix + "class_embed.bias", (num_classes + 1,))]
    for j in range(3):
        s += [(f"{prefix}mask_embed.layers.{j}.weight", (C, C)), (f"{prefix}mask_embed.layers.{j}.bias", (C,))]
    return s


def clip_visual_spec(prefix="clip_adapter.clip_model.visual.", width=768, layers=12, heads=12, patch=16, resolution=224,
                     embed_dim=512, mask_prompt_depth=0):
    """mask_prompt_depth > 0: the mask-adapted tower's `mask_embedding` [depth, G*G, width] (mask_adapted_clip/model.py:325)."""
    G = resolution // patch
    s = [(prefix + "conv1.weight", (width, 3, patch, patch)), (prefix + "class_embedding", (width,)),
         (prefix + "positional_embedding", (G * G + 1, width)), (prefix + "ln_pre.weight", (width,)), (prefix + "ln_pre.bias", (width,)),
         (prefix + "ln_post.weight", (width,)), (prefix + "ln_post.bias", (width,)), (prefix + "proj", (width, embed_dim))]
    for i in range(layers):
        p = f"{prefix}transformer.resblocks.{i}."
        s += [(p + "attn.in_proj_weight", (3 * width, width)), (p + "attn.in_proj_bias", (3 * width,)),
              (p + "attn.out_proj.weight", (width, width)), (p + "attn.out_proj.bias", (width,)),
              (p + "ln_1.weight", (width,)), (p + "ln_1.bias", (width,)), (p + "ln_2.weight", (width,)), (p + "ln_2.bias", (width,)),
              (p + "mlp.c_fc.weight", (4 * width, width)), (p + "mlp.c_fc.bias", (4 * width,)),
              (p + "mlp.c_proj.weight", (width, 4 * width)), (p + "mlp.c_proj.bias", (width,))]
    if mask_prompt_depth > 0:
        s.append((prefix + "mask_embedding", (mask_prompt_depth, G * G, width)))
    return s


def side_adapter_spec(prefix="clip_adapter.", out_dims=256, n_merge=3, **arch):
    arch = dict(arch or _CLIP_ARCH["ViT-B/16"])
    s = clip_visual_spec(prefix + "clip_model.visual.", **arch)
    s.append((prefix + "clip_model.logit_scale", ()))
    for i in range(n_merge):
        s += [(f"{prefix}attn_projs.{i}.weight", (out_dims, arch["width"], 1, 1)), (f"{prefix}attn_projs.{i}.bias", (out_dims,))]
    s.append((prefix + "bg_embed", (1, arch["embed_dim"])))
    return s


def side_decoder_spec(prefix="sem_seg_head.predictor.", C=256, clip_heads=12, Q=100):
    s = [kv for kv in video_decoder_spec(prefix, C=C, Q=Q) if "class_embed" not in kv[0]]
    for j in range(3):
        s += [(f"{prefix}attn_embed.layers.{j}.weight", (C, C)), (f"{prefix}attn_embed.layers.{j}.bias", (C,))]
    for j, co in enumerate((C, C, C * clip_heads)):
        s += [(f"{prefix}attn_mlp.layers.{j}.weight", (co, C, 1, 1)), (f"{prefix}attn_mlp.layers.{j}.bias", (co,))]
    return s


def resampler_spec(prefix="resampler.", C=256, layers=6, ffn=2048):
    s = []
    for i in range(layers):
        p = f"{prefix}long_aggregate_layers.{i}."
        s += [(p + "self_attn.in_proj_weight", (3 * C, C)), (p + "self_attn.in_proj_bias", (3 * C,)),
              (p + "self_attn.out_proj.weight", (C, C)), (p + "self_attn.out_proj.bias", (C,)),
              (p + "norm.weight", (C,)), (p + "norm.bias", (C,))]
        p = f"{prefix}short_aggregate_layers.{i}."
        s += [(p + "0.weight", (C, C, 5)), (p + "0.bias", (C,)), (p + "2.weight", (C, C, 3)), (p + "2.bias", (C,))]
        s += [(f"{prefix}aggregate_norms.{i}.weight", (C,)), (f"{prefix}aggregate_norms.{i}.bias", (C,))]
        p = f"{prefix}transformer_ffn_layers.{i}."
        s += [(p + "linear1.weight", (ffn, C)), (p + "linear1.bias", (ffn,)), (p + "linear2.weight", (C, ffn)),
              (p + "linear2.bias", (C,)), (p + "norm.weight", (C,)), (p + "norm.bias", (C,))]
    s += [(prefix + "decode_norm.weight", (C,)), (prefix + "decode_norm.bias", (C,))]
    for nm in ("attn_embed", "mask_embed"):
        for j in range(3):
            s += [(f"{prefix}{nm}.layers.{j}.weight", (C, C)), (f"{prefix}{nm}.layers.{j}.bias", (C,))]
    return s


def san_r50_spec(clip_arch=None, num_queries=100):
    arch = dict(clip_arch or _CLIP_ARCH["ViT-B/16"])
    return (resnet50_spec() + pixel_decoder_spec() + side_decoder_spec(clip_heads=arch["width"] // 64, Q=num_queries) +
            side_adapter_spec(**arch))


def brivis_r50_spec(clip_arch=None, num_queries=100):
    return san_r50_spec(clip_arch, num_queries) + resampler_spec()


def openvis_r50_spec(clip_arch=None, num_queries=100):
    arch = dict(clip_arch or _CLIP_ARCH["ViT-B/16"])
    return resnet50_spec() + pixel_decoder_spec() + video_decoder_spec(Q=num_queries) + clip_visual_spec(**arch)


def _backbone(backbone):
    """-> (backbone spec, res2..res5 channels) for "r50" or a SWIN_ARCH name."""
    if backbone == "r50":
        return resnet50_spec(), (256, 512, 1024, 2048)
    a = backbone if isinstance(backbone, dict) else SWIN_ARCH[backbone]
    return swin_spec(**a), tuple(a["embed_dim"] * 2 ** i for i in range(4))


def san_spec(backbone="r50", clip_arch=None, num_queries=100):
    arch = dict(clip_arch or _CLIP_ARCH["ViT-B/16"])
    bb, ch = _backbone(backbone)
    return (bb + pixel_decoder_spec(in_channels=ch) + side_decoder_spec(clip_heads=arch["width"] // 64, Q=num_queries) +
            side_adapter_spec(**arch))


def brivis_spec(backbone="r50", clip_arch=None, num_queries=100):
    return san_spec(backbone, clip_arch, num_queries) + resampler_spec()


def openvis_spec(backbone="r50", clip_arch=None, num_queries=100, adapter="ClipAdapter", mask_prompt_depth=3):
    """adapter: the CLIP_ADAPTER.NAME — Adapted* towers carry mask_embedding, Bg* adapters a non_object_embedding."""
    arch = dict(clip_arch or _CLIP_ARCH["ViT-B/16"])
    bb, ch = _backbone(backbone)
    depth = mask_prompt_depth if "Adapted" in adapter else 0
    s = bb + pixel_decoder_spec(in_channels=ch) + video_decoder_spec(Q=num_queries) + clip_visual_spec(**arch, mask_prompt_depth=depth)
    if adapter.startswith("Bg"):
        s.append(("clip_adapter.non_object_embedding", (1, arch["embed_dim"])))
    return s


def clip_text_spec(prefix="clip_adapter.clip_model.", width=512, layers=12, embed_dim=512, vocab=49408, context=77):
    """Text side of the CLIP state dict (mask_adapted_clip/model.py:408-423); ViT-B/16: width 512, 8 heads, embed 512;
    ViT-L/14: width 768, 12 heads, embed 768."""
    s = [(prefix + "token_embedding.weight", (vocab, width)), (prefix + "positional_embedding", (context, width)),
         (prefix + "ln_final.weight", (width,)), (prefix + "ln_final.bias", (width,)), (prefix + "text_projection", (width, embed_dim))]
    for i in range(layers):
        p = f"{prefix}transformer.resblocks.{i}."
        s += [(p + "attn.in_proj_weight", (3 * width, width)), (p + "attn.in_proj_bias", (3 * width,)),
              (p + "attn.out_proj.weight", (width, width)), (p + "attn.out_proj.bias", (width,)),
              (p + "ln_1.weight", (width,)), (p + "ln_1.bias", (width,)), (p + "ln_2.weight", (width,)), (p + "ln_2.bias", (width,)),
              (p + "mlp.c_fc.weight", (4 * width, width)), (p + "mlp.c_fc.bias", (4 * width,)),
              (p + "mlp.c_proj.weight", (width, 4 * width)), (p + "mlp.c_proj.bias", (width,))]
    return s


def spec_for_cfg(cfg):
    """Weight spec of the architecture a config names (meta-architecture, backbone, CLIP tower, queries)."""
    clip = _CLIP_ARCH[cfg.MODEL.CLIP_ADAPTER.CLIP_MODEL_NAME]
    q = cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES
    if cfg.MODEL.BACKBONE.NAME == "D2SwinTransformer":
        sw = cfg.MODEL.SWIN
        bb = dict(embed_dim=sw.EMBED_DIM, depths=tuple(sw.DEPTHS), num_heads=tuple(sw.NUM_HEADS), window=sw.WINDOW_SIZE)
    else:
        bb = "r50"
    arch = cfg.MODEL.META_ARCHITECTURE
    if arch in ("OpenVIS", "OpenVISOnline"):
        ca = cfg.MODEL.CLIP_ADAPTER
        return openvis_spec(bb, clip, q, ca.NAME, ca.MASK_PROMPT_DEPTH)
    if arch in ("SAN", "SANOnline"):
        return san_spec(bb, clip, q)
    if arch == "BriVIS":
        return brivis_spec(bb, clip, q)
    raise KeyError(f"no weight spec for META_ARCHITECTURE {arch}")


def random_init(spec, seed=42):
    """Seeded random weights with sane scales (fan-in scaled matrices, unit norms, small biases)."""
    g = torch.Generator().manual_seed(int(seed))
    sd = {}
    for key, shape in spec:
        shape = tuple(shape)
        if key.endswith("logit_scale"):
            t = torch.tensor(math.log(1 / 0.07))
        elif key.endswith("running_var"):
            t = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif key.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.1
        elif key.endswith("sampling_offsets.bias"):
            t = torch.randn(shape, generator=g) * 2.0
        elif key.endswith("sampling_offsets.weight"):
            t = torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[-1]))
        elif "query_feat" in key or "query_embed" in key:
            t = torch.randn(shape, generator=g)
        elif key.endswith("mask_embedding"):
            t = torch.randn(shape, generator=g) * 0.5             # token-sized, so that the mask prompt is visible in tests
        elif len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g) if key.endswith("weight") else 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[key] = t
    return sd


def sharpen_clip_attention(sd, gain=4.0, prefix="clip_adapter.clip_model.visual."):
    """SYNTHETIC checkpoints only (parity fixtures; never applied to a loaded checkpoint).  random_init() gives a CLIP tower whose attention
    is nearly uniform: every crop's CLS token averages its patches, the crop embeddings of one clip differ by 0.3-1 % (centred singular values
    0.03, 0.003, 0.002, ...), every query scores alike and a classification test cannot fail.  Multiplying the q and k rows of every
    `attn.in_proj_weight / in_proj_bias` of the tower by `gain` (attention logits x gain^2: the peaked attention a trained tower has) spreads
    the embeddings over many directions (gain 4: centred singular values 1.3, 0.9, 0.7, ...), so that class scores differ between queries.
    Returns a NEW dict; everything outside `prefix` is shared."""
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith(prefix) and (k.endswith("attn.in_proj_weight") or k.endswith("attn.in_proj_bias")):
            w = v.clone()
            w[: 2 * (v.shape[0] // 3)] *= gain
            out[k] = w
    return out


class _ArrayUnpickler(__import__("pickle").Unpickler):
    """Unpickler for detectron2 model-zoo `.pkl` files that resolves only what such a file needs (numpy array
    reconstruction, OrderedDict, builtin containers): a MODEL.WEIGHTS path must not be able to run arbitrary code."""
    _ALLOWED = {("collections", "OrderedDict"), ("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
                ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"),
                ("builtins", "frozenset"), ("builtins", "bytearray"), ("builtins", "complex")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise __import__("pickle").UnpicklingError(f"checkpoint pickle refers to {module}.{name}: refused")


_C2_SUFFIX = (("_bn_riv", ".norm.running_var"), ("_bn_rm", ".norm.running_mean"), ("_bn_s", ".norm.weight"),
              ("_bn_b", ".norm.bias"), ("_w", ".weight"), ("_b", ".bias"))
_C2_BRANCH = {"branch2a": "conv1", "branch2b": "conv2", "branch2c": "conv3", "branch1": "shortcut"}


def _c2_resnet_name(k):
    """Caffe2 / MSRA ResNet blob name -> detectron2 module name (detectron2 checkpoint/c2_model_loading.py
    convert_basic_c2_names, ResNet rows): conv1_w -> stem.conv1.weight, res_conv1_bn_s -> stem.conv1.norm.weight,
    res2_0_branch2a_w -> res2.0.conv1.weight, res3_0_branch1_bn_b -> res3.0.shortcut.norm.bias.  None: not a backbone blob."""
    for suf, rep in _C2_SUFFIX:
        if k.endswith(suf):
            stem, tail = k[:-len(suf)], rep
            break
    else:
        return None
    if stem in ("conv1", "res_conv1"):
        return "stem.conv1" + tail
    parts = stem.split("_")
    if len(parts) == 3 and parts[0].startswith("res") and parts[0][3:].isdigit() and parts[1].isdigit() and parts[2] in _C2_BRANCH:
        return f"{parts[0]}.{parts[1]}.{_C2_BRANCH[parts[2]]}{tail}"
    return None


def migrate_legacy_keys(sd):
    """The automatic conversions the reference applies to old checkpoints when a module's stored version is < 2, plus detectron2's
    FrozenBatchNorm2d defaults:

    * `MaskFormerHead._load_from_state_dict` (mask_former_head.py:23-45): `sem_seg_head.<x>` with <x> outside `predictor.` (and not
      already under `pixel_decoder.`) is a pixel-decoder tensor of a Mask2Former v1 checkpoint -> `sem_seg_head.pixel_decoder.<x>`;
    * `VideoMultiScaleMaskedTransformerDecoder._load_from_state_dict` (video_mask2former_transformer_decoder.py:224-245):
      `static_query` -> `query_feat` (the README.md:5 Mask2Former `.pkl`);
    * detectron2 `FrozenBatchNorm2d._load_from_state_dict` (version < 2): a norm that has `weight` but no `running_mean` /
      `running_var` (the MSRA R-50.pkl carries only `*_bn_s` / `*_bn_b`) gets zeros / ones.

    Returns (new dict, list of (old key, new key) renames); the input dict is not modified."""
    out, renamed = {}, []
    head = "sem_seg_head."
    for k, v in sd.items():
        nk = k
        if nk.startswith(head) and not nk.startswith((head + "predictor.", head + "pixel_decoder.")):
            nk = head + "pixel_decoder." + nk[len(head):]
        if "static_query" in nk:
            nk = nk.replace("static_query", "query_feat")
        if nk != k:
            renamed.append((k, nk))
        out[nk] = v
    for k in [k for k in out if k.startswith("backbone.") and k.endswith(".norm.weight")]:
        base = k[:-len("weight")]
        if base + "running_mean" not in out:
            out[base + "running_mean"] = torch.zeros_like(out[k])
        if base + "running_var" not in out:
            out[base + "running_var"] = torch.ones_like(out[k])
    return out, renamed


def load_checkpoint(path):
    """State dict of a reference checkpoint: a torch `.pth` / `.pt` file ({"model": state_dict} as DetectionCheckpointer
    writes it, or a bare state dict; loaded with weights_only=True) or a detectron2 model-zoo `.pkl` (pickle of
    {"model": {key: numpy array}, "matching_heuristics": True, ...}; read by an unpickler restricted to array data).
    A backbone-only pickle (MODEL.WEIGHTS: detectron2://ImageNetPretrained/MSRA/R-50.pkl in the reference's Base.yaml) has
    un-prefixed Caffe2 or detectron2 names, which DetectionCheckpointer resolves with its matching heuristics: here the
    Caffe2 ResNet names are converted and every backbone key gets the `backbone.` prefix of the meta-architectures."""
    if path.endswith(".pkl"):
        import numpy as np
        with open(path, "rb") as f:
            data = _ArrayUnpickler(f, encoding="latin1").load()
        model = data.get("model", data) if isinstance(data, dict) else data
        out = {}
        heur = isinstance(data, dict) and bool(data.get("matching_heuristics", False))
        for k, v in model.items():
            if k.startswith("__"):
                continue
            t = torch.from_numpy(np.ascontiguousarray(v)) if not torch.is_tensor(v) else v
            if not k.startswith(("backbone.", "sem_seg_head.", "clip_adapter.", "resampler.", "brownian_criterion.")):
                c2 = _c2_resnet_name(k)
                if c2 is not None:
                    k = "backbone." + c2
                elif (heur and "." in k) or k.startswith(("stem.", "res2.", "res3.", "res4.", "res5.")):
                    k = "backbone." + k
                else:
                    continue                      # fc1000 / optimizer blobs of a classification checkpoint
            out[k] = t
        return _migrated(out)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    return _migrated(ck.get("model", ck) if isinstance(ck, dict) else ck)


def _migrated(sd):
    sd, renamed = migrate_legacy_keys(sd)
    if renamed:                                   # the reference warns once per module (mask_former_head.py:41-45)
        import logging
        logging.getLogger(__name__).warning("old checkpoint format: %d keys converted automatically (e.g. %s -> %s)",
                                            len(renamed), renamed[0][0], renamed[0][1])
    return sd
