"""TEST INFRASTRUCTURE ONLY (build container only; never used by the product path).

Import shims that let the *reference's own Python modules* under /root/reference
be imported in this container (detectron2 / fvcore / clip / torchvision / timm are
not installed).  Used only by ``oracle/make_golden.py`` to (a) validate the CPU
restatements in ``oracle/`` against the real reference code and (b) generate the
golden vectors committed under ``tests/golden/``.

/root/reference does not exist on the GPU box; nothing in ``tests -m gpu``,
``bench.py`` or ``__graft_entry__.smoke()`` imports this file.

The stubs restate only *plumbing* (registries, decorators, thin nn wrappers) of
third-party packages the reference imports; see SURVEY.md Appendix C.
"""
import sys
import types
import importlib
import dataclasses

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

import torch
from torch import nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _Registry(dict):
    def __init__(self, name="registry"):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def _configurable(init_func=None, *, from_config=None):
    # explicit-kwargs construction only (from_config bypassed)
    if init_func is not None:
        return init_func
    return lambda f: f


class _Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d: conv -> optional norm -> optional activation."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    raise NotImplementedError(norm)


@dataclasses.dataclass
class _ShapeSpec:
    channels: int = None
    height: int = None
    width: int = None
    stride: int = None


def _c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class _Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def forward(self, x):
        return (x - self.mean.to(x)) / self.std.to(x)


_installed = False


def install():
    """Register all stub modules + synthetic reference packages (idempotent)."""
    global _installed
    if _installed:
        return
    _installed = True

    _mod("MultiScaleDeformableAttention")  # empty: forces MSDeformAttn's torch fallback

    _mod("detectron2")
    _mod("detectron2.config", configurable=_configurable)
    _mod("detectron2.layers", Conv2d=_Conv2d, ShapeSpec=_ShapeSpec, get_norm=_get_norm)
    _mod("detectron2.modeling",
         SEM_SEG_HEADS_REGISTRY=_Registry("SEM_SEG_HEADS"),
         META_ARCH_REGISTRY=_Registry("META_ARCH"),
         BACKBONE_REGISTRY=_Registry("BACKBONE"),
         build_backbone=None, build_sem_seg_head=None)
    _mod("detectron2.modeling.backbone", Backbone=nn.Module)
    sys.modules["detectron2.modeling"].Backbone = nn.Module
    sys.modules["detectron2.modeling"].ShapeSpec = _ShapeSpec
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("detectron2.utils.comm", get_local_rank=lambda: 0, synchronize=lambda: None,
         get_world_size=lambda: 1, get_rank=lambda: 0)
    _mod("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    _mod("detectron2.projects")
    _mod("detectron2.projects.point_rend")
    _mod("detectron2.projects.point_rend.point_features", get_uncertain_point_coords_with_randomness=None,
         point_sample=None)                                  # training-only helpers (criterion / matcher)
    _mod("detectron2.structures", ImageList=None, BitMasks=None, Boxes=None, Instances=None)
    _mod("detectron2.data", MetadataCatalog=None)

    _mod("fvcore")
    _mod("fvcore.nn")
    _mod("fvcore.nn.weight_init", c2_xavier_fill=_c2_xavier_fill)

    _mod("torchvision")
    _mod("torchvision.transforms")
    _mod("torchvision.transforms.transforms", Normalize=_Normalize)
    _mod("torchvision.ops", roi_align=None)

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
         trunc_normal_=nn.init.trunc_normal_)

    # vendored CLIP (a superset of openai/CLIP's model.py) stands in for `clip`
    mac = _mod("mask_adapted_clip")
    mac.__path__ = [REF_ROOT + "/third_parties/mask_adapted_clip/mask_adapted_clip"]
    mac.__package__ = "mask_adapted_clip"
    mac_model = importlib.import_module("mask_adapted_clip.model")
    clip_mod = _mod("clip", tokenize=None, load=None)
    _mod("clip.model", CLIP=mac_model.CLIP, VisionTransformer=mac_model.VisionTransformer,
         ResidualAttentionBlock=mac_model.ResidualAttentionBlock, Transformer=mac_model.Transformer,
         LayerNorm=mac_model.LayerNorm, QuickGELU=mac_model.QuickGELU)

    # synthetic packages so the reference's catch-all __init__.py files are not executed
    base = REF_ROOT + "/openvis"
    for name, path in [
        ("openvis", base),
        ("openvis.utils", base + "/utils"),
        ("openvis.modeling", base + "/modeling"),
        ("openvis.modeling.pixel_decoder", base + "/modeling/pixel_decoder"),
        ("openvis.modeling.transformer_decoder", base + "/modeling/transformer_decoder"),
        ("openvis.modeling.clip_adapter", base + "/modeling/clip_adapter"),
        ("openvis.modeling.backbone", base + "/modeling/backbone"),
    ]:
        m = _mod(name)
        m.__path__ = [path]
        m.__package__ = name


def ref(modname):
    """Import a reference module, e.g. ref('openvis.modeling.pixel_decoder.msdeformattn')."""
    install()
    return importlib.import_module(modname)
