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
    # extension keys of the CLIP tower: defaults and override syntax
    assert cfg.MODEL.CLIP_ADAPTER.RESIDUAL_STREAM == "fp16" and cfg.MODEL.CLIP_ADAPTER.FOLD_LAYERNORM is True
    cfg.merge_from_list(["MODEL.CLIP_ADAPTER.FOLD_LAYERNORM", "False"])
    assert cfg.MODEL.CLIP_ADAPTER.FOLD_LAYERNORM is False


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
              "SideAdapterFrameMultiScaleMaskedTransformerDecoder", "SideAdapterVideoMultiScaleMaskedTransformerDecoder",
              "EmbeddingVideoMultiScaleMaskedTransformerDecoder", "ProposalVideoMultiScaleMaskedTransformerDecoder",
              "EmbeddingFrameMultiScaleMaskedTransformerDecoder", "ProposalFrameMultiScaleMaskedTransformerDecoder"):
        assert TRANSFORMER_DECODER_REGISTRY.get(n) is not None
    cfg = config.get_cfg()
    emb = TRANSFORMER_DECODER_REGISTRY.get("EmbeddingFrameMultiScaleMaskedTransformerDecoder").from_config(cfg, 256, True)
    assert emb.clip_dims == cfg.MODEL.CLIP_ADAPTER.CLIP_EMBED_DIMS and emb.mask_classification       # frame decoder:157-193
    assert TRANSFORMER_DECODER_REGISTRY.get("ProposalVideoMultiScaleMaskedTransformerDecoder").from_config(cfg, 256, False).mask_classification is False


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
    sd = {"backbone.stem.conv1.weight": torch.randn(4, 3, 7, 7), "sem_seg_head.pixel_decoder.x.bias": torch.randn(5)}
    torch.save({"model": sd, "iteration": 3}, tmp_path / "m.pth")
    with open(tmp_path / "m.pkl", "wb") as f:
        pickle.dump({"model": {k: v.numpy() for k, v in sd.items()}, "__author__": "x", "matching_heuristics": True}, f)
    for name in ("m.pth", "m.pkl"):
        got = weights.load_checkpoint(str(tmp_path / name))
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_backbone_only_pickles_get_the_backbone_prefix(tmp_path):
    """detectron2 model-zoo ImageNet pickles (Base.yaml MODEL.WEIGHTS: .../MSRA/R-50.pkl) carry Caffe2 blob names or
    un-prefixed detectron2 names; DetectionCheckpointer's matching heuristics map them onto `backbone.*`."""
    import pickle
    import numpy as np
    import torch
    c2 = {"conv1_w": np.zeros((64, 3, 7, 7), np.float32), "res_conv1_bn_s": np.ones(64, np.float32),
          "res2_0_branch2a_w": np.zeros((64, 64, 1, 1), np.float32), "res2_0_branch2a_bn_b": np.zeros(64, np.float32),
          "res3_0_branch1_w": np.zeros((512, 256, 1, 1), np.float32), "res3_0_branch1_bn_rm": np.zeros(512, np.float32),
          "res5_2_branch2c_bn_riv": np.ones(2048, np.float32), "fc1000_w": np.zeros((1000, 2048), np.float32),
          "fc1000_b": np.zeros(1000, np.float32)}
    with open(tmp_path / "c2.pkl", "wb") as f:
        pickle.dump({"model": c2, "__author__": "MSRA", "matching_heuristics": True}, f)
    got = weights.load_checkpoint(str(tmp_path / "c2.pkl"))
    assert set(got) == {"backbone.stem.conv1.weight", "backbone.stem.conv1.norm.weight", "backbone.res2.0.conv1.weight",
                        "backbone.res2.0.conv1.norm.bias", "backbone.res3.0.shortcut.weight",
                        "backbone.res3.0.shortcut.norm.running_mean", "backbone.res5.2.conv3.norm.running_var",
                        # FrozenBatchNorm2d defaults for a norm that arrives without statistics (below)
                        "backbone.stem.conv1.norm.running_mean", "backbone.stem.conv1.norm.running_var"}
    d2 = {"stem.conv1.weight": np.zeros((64, 3, 7, 7), np.float32), "res4.5.conv2.norm.bias": np.zeros(256, np.float32)}
    with open(tmp_path / "d2.pkl", "wb") as f:
        pickle.dump({"model": d2, "matching_heuristics": True}, f)
    assert set(weights.load_checkpoint(str(tmp_path / "d2.pkl"))) == {"backbone." + k for k in d2}
    r50 = {k for k, _ in weights.openvis_spec("r50", dict(width=64, layers=1, heads=1, patch=16, resolution=32, embed_dim=16), 100)}
    assert {k for k in got} <= r50                             # every converted name is a key of the R50 models


def test_msra_r50_pickle_without_bn_statistics_loads_like_detectron2(tmp_path):
    """The real MSRA R-50.pkl carries only `*_bn_s` / `*_bn_b` per norm: detectron2's FrozenBatchNorm2d._load_from_state_dict
    (version < 2) fills running_mean = 0 and running_var = 1.  The converted dict must feed the ResNet weight folding."""
    import pickle
    import numpy as np
    import torch
    from openvis_amd.modeling.backbone import resnet
    rng = np.random.default_rng(0)
    c2 = {"conv1_w": rng.standard_normal((64, 3, 7, 7)).astype(np.float32), "res_conv1_bn_s": rng.random(64).astype(np.float32) + 0.5,
          "res_conv1_bn_b": rng.standard_normal(64).astype(np.float32)}
    with open(tmp_path / "R-50.pkl", "wb") as f:
        pickle.dump({"model": c2, "__author__": "MSRA", "matching_heuristics": True}, f)
    got = weights.load_checkpoint(str(tmp_path / "R-50.pkl"))
    assert torch.equal(got["backbone.stem.conv1.norm.running_mean"], torch.zeros(64))
    assert torch.equal(got["backbone.stem.conv1.norm.running_var"], torch.ones(64))
    w, b = resnet._fold(got, "backbone.stem.conv1")                       # KeyError before the defaults existed
    scale = torch.from_numpy(c2["res_conv1_bn_s"]) * (1.0 + 1e-5) ** -0.5
    assert torch.allclose(w, (torch.from_numpy(c2["conv1_w"]) * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1))
    assert torch.allclose(b, torch.from_numpy(c2["res_conv1_bn_b"]))


def test_legacy_checkpoint_keys_are_migrated(tmp_path):
    """mask_former_head.py:23-45 (`sem_seg_head.* -> sem_seg_head.pixel_decoder.*` unless under `predictor.`) and
    video_mask2former_transformer_decoder.py:224-245 (`static_query -> query_feat`): a Mask2Former v1 checkpoint (README.md:5)."""
    import torch
    old = {"sem_seg_head.adapter_1.weight": torch.randn(4, 4, 1, 1), "sem_seg_head.layer_1.norm.bias": torch.randn(4),
           "sem_seg_head.predictor.static_query.weight": torch.randn(10, 4), "sem_seg_head.predictor.query_embed.weight": torch.randn(10, 4),
           "sem_seg_head.pixel_decoder.mask_features.bias": torch.randn(4), "backbone.res2.0.conv1.weight": torch.randn(4, 4, 1, 1)}
    new, renamed = weights.migrate_legacy_keys(old)
    assert set(new) == {"sem_seg_head.pixel_decoder.adapter_1.weight", "sem_seg_head.pixel_decoder.layer_1.norm.bias",
                        "sem_seg_head.predictor.query_feat.weight", "sem_seg_head.predictor.query_embed.weight",
                        "sem_seg_head.pixel_decoder.mask_features.bias", "backbone.res2.0.conv1.weight"}
    assert len(renamed) == 3 and torch.equal(new["sem_seg_head.predictor.query_feat.weight"], old["sem_seg_head.predictor.static_query.weight"])
    assert set(old) != set(new) and "sem_seg_head.adapter_1.weight" in old               # the input dict is left alone
    # idempotent, and a current-format dict passes through unchanged
    again, renamed2 = weights.migrate_legacy_keys(new)
    assert set(again) == set(new) and not renamed2
    torch.save({"model": old}, tmp_path / "v1.pth")
    assert set(weights.load_checkpoint(str(tmp_path / "v1.pth"))) == set(new)
    # a migrated v1 dict has every key the current spec asks for under those names
    spec = {k for k, _ in weights.openvis_spec("r50", dict(width=64, layers=1, heads=1, patch=16, resolution=32, embed_dim=16), 100)}
    assert "sem_seg_head.predictor.query_feat.weight" in spec and "sem_seg_head.pixel_decoder.adapter_1.weight" in spec


def test_checkpoint_files_cannot_execute_code(tmp_path):
    """MODEL.WEIGHTS is user-supplied: a pickle that refers to anything but array data is refused (.pkl), torch.load
    runs with weights_only=True (.pth)."""
    import os
    import pickle
    import pytest
    import torch

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    with open(tmp_path / "evil.pkl", "wb") as f:
        pickle.dump({"model": {"x": Evil()}}, f)
    with pytest.raises(pickle.UnpicklingError):
        weights.load_checkpoint(str(tmp_path / "evil.pkl"))
    torch.save({"model": {"x": Evil()}}, tmp_path / "evil.pth")
    with pytest.raises(Exception):
        weights.load_checkpoint(str(tmp_path / "evil.pth"))


def test_evaluator_handoff_formats():
    """instances_to_coco_json_video / instances_to_burst_json_video (ytvis_eval.py:258-301, burst_eval.py:177-240) from
    dense masks and from pre-encoded RLE give the same records; BURST keeps a track only on frames with > 20 pixels."""
    import numpy as np
    import torch
    from openvis_amd import evals, rle
    T, H, W = 3, 12, 17
    rng = np.random.default_rng(0)
    m0 = rng.random((T, H, W)) > 0.5
    m1 = np.zeros((T, H, W), bool)
    m1[0, 2:6, 3:9] = True            # 24 px: kept
    m1[1, 0:4, 0:5] = True            # 20 px: dropped (needs > 20)
    outputs = {"pred_scores": [0.9, 0.4], "pred_labels": [5, 2], "pred_entropys": [0.1, 0.7],
               "pred_masks": [torch.from_numpy(m0), torch.from_numpy(m1)]}
    inputs = [{"video_id": 7, "length": T, "width": W, "height": H, "seq_name": "s", "dataset": "d",
               "annotated_image_paths": ["a", "b", "c"]}]
    coco = evals.instances_to_coco_json_video(inputs, outputs)
    assert [r["category_id"] for r in coco] == [5, 2] and coco[1]["entropy"] == 0.7 and coco[0]["video_id"] == 7
    for r, m in zip(coco, (m0, m1)):
        for seg, f in zip(r["segmentations"], m):
            assert seg["size"] == [H, W]
            assert (rle.counts_to_mask(rle.string_to_counts(seg["counts"]), H, W) == f).all()
    pre = dict(outputs, pred_masks_rle=[r["segmentations"] for r in coco])
    del pre["pred_masks"]
    assert evals.instances_to_coco_json_video(inputs, pre) == coco
    (burst,) = evals.instances_to_burst_json_video(inputs, pre)
    assert burst["seq_name"] == "s" and burst["annotated_image_paths"] == ["a", "b", "c"] and len(burst["segmentations"]) == T
    assert burst["track_category_ids"] == {0: 5, 1: 2}
    assert [sorted(f) for f in burst["segmentations"]] == [[0, 1], [0], [0]]
    a = burst["segmentations"][0][1]
    assert a == {"rle": coco[1]["segmentations"][0]["counts"], "is_gt": False, "score": 0.4, "entropy": 0.7}


def test_mask_former_head_dispatch_mirrors_the_reference_branches():
    """mask_former_head.py:119-135: TRANSFORMER_IN_FEATURE selects what the decoder is fed.  Every reference config uses
    "multi_scale_pixel_decoder"; the single-map branches reach *MultiScale* decoders that assert three feature levels and fail in the
    reference too -- here they fail the same way instead of being refused at construction (round 5's review, missing item 5)."""
    import pytest
    from openvis_amd.modeling.mask_former_head import MaskFormerHead

    class PD:
        def forward_features(self, features, extra=None):
            return "mask_features", "enc0", ["ms0", "ms1", "ms2"]

    class Dec:
        num_feature_levels = 3

        def __call__(self, x, mask_features, mask=None):
            return {"x": x, "mask_features": mask_features, "mask": mask}

    shape = {k: dict(channels=c, stride=s) for k, c, s in (("res2", 256, 4), ("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    mk = lambda tif: MaskFormerHead(shape, num_classes=1, pixel_decoder=PD(), transformer_predictor=Dec(), transformer_in_feature=tif)
    out = mk("multi_scale_pixel_decoder")({"res5": "f5"}, mask="m")
    assert out == {"x": ["ms0", "ms1", "ms2"], "mask_features": "mask_features", "mask": "m"}
    for tif in ("transformer_encoder", "pixel_embedding", "res5"):
        with pytest.raises(AssertionError):
            mk(tif)({"res5": "f5"})
    with pytest.raises(TypeError):
        mk("side_adapter")({"res5": "f5"})
