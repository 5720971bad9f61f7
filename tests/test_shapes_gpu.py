"""GPU: shape robustness -- every meta-architecture runs on frame sizes whose strided maps are odd / not multiples of the
kernels' vector widths (alignment and tail handling), and returns well-formed output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [(360, 640), (333, 517), (250, 190), (97, 131)]
ARCHS = [("OpenVIS", "VideoMultiScaleMaskedTransformerDecoder"), ("OpenVISOnline", "FrameMultiScaleMaskedTransformerDecoder"),
         ("SAN", "SideAdapterVideoMultiScaleMaskedTransformerDecoder"), ("SANOnline", "SideAdapterFrameMultiScaleMaskedTransformerDecoder"),
         ("BriVIS", "SideAdapterFrameMultiScaleMaskedTransformerDecoder")]


@pytest.mark.parametrize("arch,decoder", ARCHS)
def test_all_archs_run_on_odd_frame_sizes(arch, decoder):
    import bench
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from openvis_amd.modeling.clip_adapter.side_adapter import SideAdapter
    from tests.test_openvis_gpu import CLIP_ARCH
    from tests.test_san_gpu import SAN_E2E_ARCH

    K = 11
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_shapes").set(thing_classes=names)
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = arch
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = decoder
    cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS = 4
    model = config.build_model(cfg)
    if arch.startswith("OpenVIS"):
        sd = weights.random_init(weights.openvis_spec("r50", CLIP_ARCH, 100), seed=3)
        model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision="fp16")
        dim = CLIP_ARCH["embed_dim"]
    else:
        sd = weights.random_init(weights.brivis_spec("r50", SAN_E2E_ARCH, 100), seed=3)
        model.clip_adapter = SideAdapter("tiny", broken_idx=3, merge_ids=[1, 2, 3], num_queries=100, arch=SAN_E2E_ARCH, precision="fp16")
        dim = SAN_E2E_ARCH["embed_dim"]
    model.load_state_dict(sd)
    model.clip_adapter.set_text_features(names, bench.synth_text(K, dim))
    for (H, W) in SIZES:
        frames = bench.synth_frames(3, H, W, H + W, "cpu")
        out = model([{"image": [f for f in frames], "dataset_name": "synthetic_shapes"}])
        assert len(out["pred_masks"]) == 10 and tuple(out["pred_masks"][0].shape) == (3, H, W), (H, W)
        assert len(out["pred_scores"]) == 10 and all(np.isfinite(s) and 0.0 <= s <= 1.0 for s in out["pred_scores"]), (H, W)
        assert all(0 <= l < K for l in out["pred_labels"])


def test_no_valid_mask_gives_the_reference_empty_output():
    """openvis.py:127-128 / video_maskformer.py:262-266: when no query has a positive mask pixel there is nothing to
    classify and the output lists are empty."""
    import bench
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from tests.test_openvis_gpu import CLIP_ARCH
    names = [f"class_{i}" for i in range(5)]
    MetadataCatalog.get("synthetic_empty").set(thing_classes=names)
    model = config.build_model(config.get_cfg())
    model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision="fp16")
    model.load_state_dict(weights.random_init(weights.openvis_spec("r50", CLIP_ARCH, 100), seed=3))
    model.clip_adapter.set_text_features(names, bench.synth_text(5, CLIP_ARCH["embed_dim"]))
    frames = bench.synth_frames(2, 96, 128, 0, "cuda")
    masks = torch.full((100, 2, 24, 32), -10.0, device="cuda")
    probs, row_ids, extras = model.open_vocabulary_inference(torch.zeros(100, 2, device="cuda"), masks, frames, names, (96, 128))
    assert probs is None and row_ids is None and not extras["valid"].any()
    out = model.inference_video(100, 5, probs, row_ids, masks, (96, 128), (96, 128), 96, 128)
    assert out["pred_masks"] == [] and out["pred_scores"] == [] and out["pred_labels"] == []
