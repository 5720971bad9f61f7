from .adapter import AdaptedClipAdapter, BgAdaptedClipAdapter, BgClipAdapter, ClipAdapter  # noqa: F401

ADAPTER_REGISTER = {x.__name__: x for x in [ClipAdapter, BgClipAdapter, AdaptedClipAdapter, BgAdaptedClipAdapter]}


def build_clip_adapter(cfg):
    """openvis/modeling/clip_adapter/__init__.py:9-15.  (SideAdapter is built by the SAN / BriVIS meta-architectures
    themselves: openvis_amd/modeling/clip_adapter/side_adapter.py.)"""
    precision = cfg.get("PRECISION", "fp16")
    stream = cfg.get("RESIDUAL_STREAM", "fp16")
    if stream not in ("fp16", "fp32"):
        raise ValueError(f"MODEL.CLIP_ADAPTER.RESIDUAL_STREAM must be 'fp16' or 'fp32', got {stream!r}")
    if cfg.NAME in ("ClipAdapter", "BgClipAdapter"):
        ad = ADAPTER_REGISTER[cfg.NAME](cfg.CLIP_MODEL_NAME, text_templates=cfg.PROMPT_NAME, precision=precision)
    elif cfg.NAME in ("AdaptedClipAdapter", "BgAdaptedClipAdapter"):
        ad = ADAPTER_REGISTER[cfg.NAME](cfg.CLIP_MODEL_NAME, mask_prompt_depth=cfg.MASK_PROMPT_DEPTH,
                                        mask_prompt_fwd=cfg.MASK_PROMPT_FWD, text_templates=cfg.PROMPT_NAME,
                                        precision=precision)
    else:
        raise NotImplementedError(f"clip adapter {cfg.NAME} is not in ADAPTER_REGISTER {sorted(ADAPTER_REGISTER)}")
    ad.visual.stream16 = precision == "fp16" and stream == "fp16"
    ad.visual.fold_ln = bool(cfg.get("FOLD_LAYERNORM", True))
    ad.crop_list = cfg.get("CROP_LIST", "auto")
    if ad.crop_list not in ("auto", "host", "device"):
        raise ValueError(f"MODEL.CLIP_ADAPTER.CROP_LIST must be 'auto', 'host' or 'device', got {ad.crop_list!r}")
    return ad
