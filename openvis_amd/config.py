"""Config surface of the reference (openvis/config.py:6-166 + detectron2 defaults used by the path), as a small
yacs-like CfgNode with `_BASE_` yaml inheritance and KEY VALUE overrides (train_net.py:256-282).  Only keys the
inference path reads get defaults; unknown yaml keys are accepted so the reference's yaml files load unchanged."""
import copy
import os

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(_to_node(v))
            else:
                self[k] = _to_node(v)
        return self

    def merge_from_file(self, path):
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        base = d.pop("_BASE_", None)
        if base:
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        return self.merge(d)

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
        return self


def _to_node(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        n = CfgNode()
        for k, x in v.items():
            n[k] = _to_node(x)
        return n
    return v


def get_cfg():
    """Defaults for the keys the eval path reads (detectron2 defaults + add_*_config of openvis/config.py)."""
    return _to_node({
        "MODEL": {
            "META_ARCHITECTURE": "OpenVIS", "DEVICE": "cuda", "WEIGHTS": "",
            # not a reference key — MI355X arithmetic policy of the dense path:
            #   "mixed" = the reference's own GPU policy (autocast: backbone GEMM operands fp16 with f32 accumulation;
            #             pixel decoder forced f32, msdeformattn.py:329),  "fp32" = exact f32 everywhere.
            #   The transformer decoder follows MASK_FORMER.DECODER_PRECISION below.
            "PRECISION": "mixed",
            # not reference keys -- per-stage operand dtype under "mixed" ("auto" = chosen per meta-architecture so that every
            # query mask keeps IoU >= 0.999 against the f32 oracle at 720p; tools/exp_policy_mix.py, exp_policy_mix_san.py):
            #   OpenVIS / OpenVISOnline: backbone fp16 operands (masks depend on backbone + pixel decoder + decoder only);
            #   SAN / SANOnline / BriVIS: the side adapter injects CLIP features into the pixel decoder, and fp16 operands in
            #   EITHER the backbone or the side adapter's ViT put every query below 0.999 (SANOnline 0.9866-0.9974, BriVIS
            #   0.9975-0.9987; tracks and logits are unaffected) -> backbone, side adapter and resampler run f32 operands.
            "BACKBONE_PRECISION": "auto", "RESAMPLER_PRECISION": "auto",
            # not a reference key -- how a LARGE f32 GEMM/conv is put on the bf16 MFMA (csrc/gemm_f32x3.h; process-wide, applied
            # by build_model):
            #   "bf16x3" = six products of the exact 3-way bf16 split, f32-grade (rel. err ~1e-7);
            #   "bf16x2" = three products of the two leading planes: 16 significand bits per operand (rel. err < 2^-16, 64 x
            #              finer than autocast's fp16 operands, 32 x finer than the TF32 the reference's f32 convolutions get
            #              from cuDNN by default), products and sums in f32, half the MFMA work;
            #   "f32"    = the native f32 MFMA (an fmaf chain) for every size;
            #   "fp16x2" = (round 4) three products hi hi + hi lo + lo hi of the fp16 split (11 + 11 significand bits per operand) on the FP16
            #              MFMA at the cost of bf16x2, for the constant-weight layers; operands are moved to the top of fp16's range by
            #              power-of-two scales (include/openvis_hip.h).  The bound is ABSOLUTE + relative, not purely relative: an
            #              activation a is carried with error <= max(2^-22 |a|, 2^-29) (a_scale = 16: below |a| ~ 1e-3 the lo plane runs
            #              into fp16's subnormal spacing 2^-24 / a_scale and a keeps 16-19 significant bits; below 4e-6 hi is subnormal
            #              too), weights with 2^-22 |w| (they sit at the top of the range).  Against a row's own scale -- the measure of
            #              tests/test_fp16x2_gpu.py, max |C - C64| / (|A||W| + |b| + |R|) -- this is the f32 grade (2.0-3.0e-7 measured,
            #              native f32 MFMA 3.6-4.5e-7); it is NOT a 2^-22 relative bound on an element whose row is all tiny.  An
            #              activation beyond 65 504 / 16 raises a device flag and the clip is repeated under bf16x3
            #              (VideoMaskFormer._range_guard); GEMMs between two activations stay on bf16x3;
            #   "auto"   = "fp16x2" under MODEL.PRECISION "mixed" (the reference keeps these layers in f32, msdeformattn.py:329: the
            #              timed policy is f32-grade), "bf16x3" under "fp32".  "bf16x2" is an explicit opt-in (16 significand bits per
            #              operand; measured at 720p against the f32 oracle, profiles/r02/bf16x2.txt: per-query mask IoU min 0.99928).
            "F32_GEMM_SPLIT": "auto",
            "PIXEL_MEAN": [123.675, 116.280, 103.530], "PIXEL_STD": [58.395, 57.120, 57.375],
            "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 0},
            "RESNETS": {"DEPTH": 50, "STRIDE_IN_1X1": False, "OUT_FEATURES": ["res2", "res3", "res4", "res5"],
                        "STEM_OUT_CHANNELS": 64, "NORM": "FrozenBN"},
            "SWIN": {"PRETRAIN_IMG_SIZE": 224, "PATCH_SIZE": 4, "EMBED_DIM": 96, "DEPTHS": [2, 2, 6, 2],
                     "NUM_HEADS": [3, 6, 12, 24], "WINDOW_SIZE": 7, "MLP_RATIO": 4.0, "QKV_BIAS": True, "QK_SCALE": None,
                     "DROP_RATE": 0.0, "ATTN_DROP_RATE": 0.0, "DROP_PATH_RATE": 0.3, "APE": False, "PATCH_NORM": True,
                     "OUT_FEATURES": ["res2", "res3", "res4", "res5"], "USE_CHECKPOINT": False},
            "SEM_SEG_HEAD": {"NAME": "MaskFormerHead", "IGNORE_VALUE": 255, "NUM_CLASSES": 1, "LOSS_WEIGHT": 1.0,
                             "CONVS_DIM": 256, "MASK_DIM": 256, "NORM": "GN",
                             "PIXEL_DECODER_NAME": "MSDeformAttnPixelDecoder",
                             "IN_FEATURES": ["res2", "res3", "res4", "res5"],
                             "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res3", "res4", "res5"],
                             "COMMON_STRIDE": 4, "TRANSFORMER_ENC_LAYERS": 6},
            "MASK_FORMER": {"TRANSFORMER_DECODER_NAME": "VideoMultiScaleMaskedTransformerDecoder",
                            "TRANSFORMER_IN_FEATURE": "multi_scale_pixel_decoder", "HIDDEN_DIM": 256,
                            "NUM_OBJECT_QUERIES": 100, "NHEADS": 8, "DROPOUT": 0.0, "DIM_FEEDFORWARD": 2048,
                            "ENC_LAYERS": 0, "DEC_LAYERS": 10, "PRE_NORM": False, "ENFORCE_INPUT_PROJ": False,
                            "SIZE_DIVISIBILITY": 32,
                            # not a reference key: GEMM operand dtype of the masked-attention decoder under
                            # MODEL.PRECISION "mixed".  "fp32" (default) although autocast would run it in fp16: the
                            # decoder is 7 % of the step, and its fp16 operands are what moves mask boundaries -- at
                            # 720p every query mask keeps IoU >= 0.9992 against the f32 oracle with "fp32", while 30 of
                            # 100 fall below 0.999 with "fp16" (tools/exp_policy_mix.py); costs 0.9 ms per clip.
                            "DECODER_PRECISION": "fp32",
                            "TEST": {"OBJECT_MASK_THRESHOLD": 0.8, "OVERLAP_THRESHOLD": 0.8, "WINDOW_INFERENCE": False,
                                     # not a reference key: return COCO RLE (encoded on the GPU) instead of dense masks
                                     "OUTPUT_RLE": False,
                                     "WINDOW_SIZE": 10}},
            "CLIP_ADAPTER": {"NAME": "ClipAdapter", "PROMPT_NAME": "vild", "CLIP_MODEL_NAME": "ViT-B/16",
                             "CLIP_NUM_HEADS": 12, "CLIP_EMBED_DIMS": 512, "MERGE_IDS": [3, 6, 9], "BROKEN_ID": 9,
                             "CLIP_ENSEMBLE": True, "CLIP_ENSEMBLE_WEIGHT": 0.8, "MASK_PROMPT_DEPTH": 3,
                             "MASK_PROMPT_FWD": True,
                             # not a reference key: GEMM operand dtype of the CLIP tower on MI355X ("fp16" as the
                             # reference's GPU CLIP, or "fp32")
                             "PRECISION": "fp16",
                             # not a reference key: dtype of the ViT's residual stream under PRECISION "fp16" for plain crops:
                             # "fp16" = what the reference's fp16 CLIP keeps between blocks (one rounding per sub-block; measured
                             # max |cos error| 9.7e-5 against 5.2e-5 with "fp32", bound 1e-3; profiles/r02/logit_bound.txt),
                             # half the residual / LayerNorm traffic of the tower
                             "RESIDUAL_STREAM": "fp16",
                             # not a reference key: with the fp16 stream, ln_1 / ln_2 are folded into in_proj / c_fc and their statistics
                             # come out of the epilogue of out_proj / c_proj (one fp16 rounding LESS than LayerNorm kernel + GEMM; -1.0 ms
                             # per 720p clip); False = LayerNorm kernels (the round-2 arithmetic)
                             "FOLD_LAYERNORM": True,
                             # not a reference key -- where ClipAdapter's crop list (adapter.py:86-102) is built:
                             #   "host"   = as the reference: the boxes are read back (one device -> host sync in the middle of the forward),
                             #              the valid (frame, query) pairs are compacted on the host and the tower runs on exactly those;
                             #   "device" = a kernel builds the list for EVERY (frame, query) (empty masks become zero images whose logits
                             #              the aggregation ignores): no read-back, no data-dependent shape, the host runs ahead of the
                             #              GPU through the whole forward -- but the tower also spends work on the empty masks;
                             #   "auto"   = "device" while the previous clips had >= 90 % non-empty masks (the count rides back with the
                             #              outputs), "host" otherwise and for the first clip.  Same results either way.
                             "CROP_LIST": "auto",
                             # SideAdapter (SAN / SANOnline / BriVIS) tower: "auto" = fp32 (see BACKBONE_PRECISION)
                             "SIDE_PRECISION": "auto"},
        },
        "INPUT": {"SAMPLING_FRAME_NUM": 2, "MIN_SIZE_TEST": 360, "MAX_SIZE_TEST": 1333, "FORMAT": "RGB"},
        "DATASETS": {"TEST": ["burst_val"]},
        "SEED": 42,
    })


def build_model(cfg):
    """detectron2.modeling.build_model: META_ARCH_REGISTRY.get(name)(cfg) via from_config."""
    from . import openvis, san, brivis  # noqa: F401  (registers the meta-architectures)
    from .registry import META_ARCH_REGISTRY
    cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    split = f32_gemm_split(cfg)
    if split not in F32_GEMM_SPLITS:
        raise ValueError(f"MODEL.F32_GEMM_SPLIT must be one of {sorted(F32_GEMM_SPLITS)}, got {split!r}")
    model = cls(**cls.from_config(cfg))
    model.f32_gemm_mode = F32_GEMM_SPLITS[split]          # applied at the start of every forward (VideoMaskFormer._frames_to_device)
    model.output_rle = bool(cfg.MODEL.MASK_FORMER.TEST.get("OUTPUT_RLE", False))
    return model


F32_GEMM_SPLITS = {"f32": 0, "bf16x3": 1, "bf16x2": 2, "fp16x2": 3}


def f32_gemm_split(cfg):
    """MODEL.F32_GEMM_SPLIT with "auto" resolved (see get_cfg)."""
    split = cfg.MODEL.get("F32_GEMM_SPLIT", "auto")
    if split == "auto":
        return "bf16x3" if cfg.MODEL.get("PRECISION", "mixed") == "fp32" else "fp16x2"
    return split


SIDE_ADAPTER_ARCHS = ("SAN", "SANOnline", "BriVIS")


def _stage_precision(cfg, value, side_default):
    if cfg.MODEL.get("PRECISION", "mixed") == "fp32":
        return "fp32"
    if value in ("fp16", "fp32"):
        return value
    return side_default if cfg.MODEL.META_ARCHITECTURE in SIDE_ADAPTER_ARCHS else "fp16"


def backbone_precision(cfg):
    """GEMM operand dtype of the backbone (MODEL.BACKBONE_PRECISION; see get_cfg)."""
    return _stage_precision(cfg, cfg.MODEL.get("BACKBONE_PRECISION", "auto"), "fp32")


def side_adapter_precision(cfg):
    """GEMM operand dtype of the SideAdapter's CLIP tower (MODEL.CLIP_ADAPTER.SIDE_PRECISION)."""
    return _stage_precision(cfg, cfg.MODEL.CLIP_ADAPTER.get("SIDE_PRECISION", "auto"), "fp32")


def resampler_precision(cfg):
    """GEMM operand dtype of BriVIS' TemporalInstanceResampler (MODEL.RESAMPLER_PRECISION)."""
    return _stage_precision(cfg, cfg.MODEL.get("RESAMPLER_PRECISION", "auto"), "fp32")


def decoder_precision(cfg):
    """GEMM operand dtype of the transformer decoder: "fp32" unless MODEL.PRECISION is "mixed" AND
    MODEL.MASK_FORMER.DECODER_PRECISION asks for the autocast behaviour ("fp16")."""
    if cfg.MODEL.get("PRECISION", "mixed") == "fp32":
        return "fp32"
    return "fp16" if cfg.MODEL.MASK_FORMER.get("DECODER_PRECISION", "fp32") == "fp16" else "fp32"
