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
