from .adapter import ClipAdapter  # noqa: F401

ADAPTER_REGISTER = {"ClipAdapter": ClipAdapter}


def build_clip_adapter(cfg):
    """openvis/modeling/clip_adapter/__init__.py:9-15 (ClipAdapter; the other adapters are later §8 rows)."""
    if cfg.NAME in ("ClipAdapter",):
        return ADAPTER_REGISTER[cfg.NAME](cfg.CLIP_MODEL_NAME, text_templates=cfg.PROMPT_NAME,
                                          precision=cfg.get("PRECISION", "fp16"))
    raise NotImplementedError(f"clip adapter {cfg.NAME} is not built yet (SURVEY.md §8a A13 / later rows)")
