"""Top-level alias so the reference's unmodified
``import MultiScaleDeformableAttention as MSDA`` (ops/functions/ms_deform_attn_func.py:22)
resolves to the MI355X implementation when this repo root is on sys.path."""
from openvis_amd.MultiScaleDeformableAttention import ms_deform_attn_forward, ms_deform_attn_backward  # noqa: F401
