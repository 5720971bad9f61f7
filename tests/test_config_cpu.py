"""CPU: the config surface (openvis_amd/config.py) loads yaml files laid out like the reference's (`_BASE_` inheritance,
KEY VALUE overrides, train_net.py:256-282), resolves every registry name the shipped configs use, and -- in the build
container, where /root/reference exists -- loads the reference's own yaml files unchanged."""
import glob
import os

import pytest

from openvis_amd import config, weights
from openvis_amd.registry import META_ARCH_REGISTRY, BACKBONE_REGISTRY, SEM_SEG_HEADS_REGISTRY, TRANSFORMER_DECODER_REGISTRY

REF_CFG = "/root/reference/configs/openvoc_ytvis_coco"


def test_yaml_base_inheritance_and_overrides(tmp_path):
    (tmp_path / "Base.yaml").write_text("MODEL:\n  META_ARCHITECTURE: \"OpenVIS\"\n  RESNETS:\n    DEPTH: 50\nINPUT:\n  MIN_SIZE_TEST: 360\n")
    (tmp_path / "child.yaml").write_text("_BASE_: Base.yaml\nMODEL:\n  META_ARCHITECTURE: \"BriVIS\"\n  MASK_FORMER:\n"
                                         "    TRANSFORMER_DECODER_NAME: \"SideAdapterFrameMultiScaleMaskedTransformerDecoder\"\n"
                                         "  CLIP_ADAPTER:\n    MERGE_IDS: [6, 12, 18]\n")
    cfg = config.get_cfg()
    cfg.merge_from_file(str(tmp_path / "child.yaml"))
    cfg.merge_from_list(["INPUT.MIN_SIZE_TEST", "480", "MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "200"])
    assert cfg.MODEL.META_ARCHITECTURE == "BriVIS" and cfg.MODEL.RESNETS.DEPTH == 50
    assert cfg.INPUT.MIN_SIZE_TEST == 480 and cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES == 200
    assert cfg.MODEL.CLIP_ADAPTER.MERGE_IDS == [6, 12, 18]


def test_registry_names_of_the_reference_surface():
    from openvis_amd import openvis, san, brivis  # noqa: F401
    import openvis_amd.modeling  # noqa: F401
    for n in ("OpenVIS", "OpenVISOnline", "SAN", "SANOnline", "BriVIS", "VideoMaskFormer", "MinVIS"):
        assert META_ARCH_REGISTRY.get(n) is not None
    for n in ("build_resnet_backbone", "D2SwinTransformer"):
        assert BACKBONE_REGISTRY.get(n) is not None
    for n in ("MaskFormerHead", "MSDeformAttnPixelDecoder"):
        assert SEM_SEG_HEADS_REGISTRY.get(n) is not None
    for n in ("VideoMultiScaleMaskedTransformerDecoder", "FrameMultiScaleMaskedTransformerDecoder",
              "SideAdapterFrameMultiScaleMaskedTransformerDecoder", "SideAdapterVideoMultiScaleMaskedTransformerDecoder"):
        assert TRANSFORMER_DECODER_REGISTRY.get(n) is not None


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs only exist in the build container")
def test_reference_yaml_files_load_unchanged():
    files = [f for f in glob.glob(REF_CFG + "/*.yaml") + glob.glob(REF_CFG + "/swin/*.yaml") if "Base" not in f and "simplebsl" not in f]
    assert len(files) >= 8
    for f in files:
        cfg = config.get_cfg()
        cfg.merge_from_file(f)
        assert META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE) is not None, f
        assert BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME) is not None, f
        assert TRANSFORMER_DECODER_REGISTRY.get(cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME) is not None, f
        spec = weights.spec_for_cfg(cfg)                      # architecture described by the yaml is one we can build
        assert len(spec) > 500


def test_load_checkpoint_pth_and_pkl(tmp_path):
    import pickle
    import numpy as np
    import torch
    sd = {"backbone.stem.conv1.weight": torch.randn(4, 3, 7, 7), "x.bias": torch.randn(5)}
    torch.save({"model": sd, "iteration": 3}, tmp_path / "m.pth")
    with open(tmp_path / "m.pkl", "wb") as f:
        pickle.dump({"model": {k: v.numpy() for k, v in sd.items()}, "__author__": "x", "matching_heuristics": True}, f)
    for name in ("m.pth", "m.pkl"):
        got = weights.load_checkpoint(str(tmp_path / name))
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
