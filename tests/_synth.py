"""Deterministic synthetic weights/inputs shared by oracle/make_golden.py and the tests.

Fixtures under tests/golden/ store only (key, shape) specs, seeds and the REFERENCE's outputs;
weights and inputs are regenerated here from the seed (same torch build on the GPU box)."""
import math

import torch


def synth_tensor(key, shape, g):
    shape = tuple(shape)
    if key.endswith("running_var"):
        return torch.rand(shape, generator=g) * 0.5 + 0.75
    if key.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("sampling_offsets.bias"):
        return torch.randn(shape, generator=g) * 2.0
    if key.endswith("sampling_offsets.weight"):
        return torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[-1]))
    if "query_feat" in key or "query_embed" in key:
        return torch.randn(shape, generator=g)
    if len(shape) == 1:
        if key.endswith("weight"):          # norm scales
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))


def synth_weights(spec, seed, prefix=""):
    """spec: iterable of (key, shape) -> {prefix+key: tensor}, generated in the given order."""
    g = torch.Generator().manual_seed(int(seed))
    return {prefix + k: synth_tensor(k, s, g) for k, s in spec}


def synth_inputs(shapes, seed, scale=1.0):
    g = torch.Generator().manual_seed(int(seed))
    return [torch.randn(tuple(s), generator=g) * scale for s in shapes]


def spec_of(state_dict):
    return [(k, tuple(v.shape)) for k, v in state_dict.items() if v.dtype.is_floating_point]


# ---- inputs of the reference-forward fixtures (oracle/make_golden_glue.py `forward`, tests/test_oracle_glue.py, tests/test_glue_gpu.py) ----
GLUE_T, GLUE_H, GLUE_W, GLUE_K, GLUE_Q = 3, 60, 90, 7, 100
GLUE_OUT_HW = (75, 112)
# the vendored CLIP's constructor arguments (mask_adapted_clip/model.py: CLIP.__init__): ViT of 4 layers, width 256 (4 heads), patch 16,
# 64 x 64 input, 64-d embedding; a one-layer text tower that the fixtures never run (the text features are given)
GLUE_CLIP = dict(embed_dim=64, image_resolution=64, vision_layers=4, vision_width=256, vision_patch_size=16, mask_prompt_depth=0,
                 context_length=8, vocab_size=64, transformer_width=64, transformer_heads=2, transformer_layers=1)


def glue_frames(seed):
    """uint8 [T,3,H,W]: noise + one moving blob (masks are not trivial)."""
    g = torch.Generator().manual_seed(int(seed))
    base = torch.rand(GLUE_T, 3, GLUE_H, GLUE_W, generator=g) * 255
    yy, xx = torch.meshgrid(torch.arange(GLUE_H), torch.arange(GLUE_W), indexing="ij")
    for t in range(GLUE_T):
        blob = 120 * torch.exp(-(((yy - 28 - 4 * t) / 13.0) ** 2 + ((xx - 45 + 6 * t) / 19.0) ** 2))
        base[t] = (base[t] * 0.4 + blob).clamp(0, 255)
    return base.to(torch.uint8)


def glue_text(seed, dim, K=GLUE_K):
    g = torch.Generator().manual_seed(int(seed))
    base = torch.randn(1, dim, generator=g)
    return torch.nn.functional.normalize(base + 0.05 * torch.randn(K, dim, generator=g), dim=-1)
