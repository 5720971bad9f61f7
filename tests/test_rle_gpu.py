"""GPU: COCO RLE of the output masks -- kernel vs the checker oracle/rle_ref.py (plain-loop restatement of
pycocotools' rleEncode / rleToString, pinned by known answers in tests/test_oracle_rle.py), string round trip, and the model's
RLE output vs its dense masks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,H,W,p", [(3, 37, 53, 0.5), (2, 720, 1280, 0.3), (4, 1, 1, 0.5), (2, 128, 128, 0.0), (2, 100, 17, 1.0),
                                     (1, 129, 127, 0.02)])
def test_rle_kernel_vs_numpy(n, H, W, p):
    from openvis_amd import ops, rle
    from oracle import rle_ref
    rng = np.random.default_rng(H * W + n)
    masks = rng.random((n, H, W)) < p
    if p == 0.3:                                          # blob-like masks (few long runs) next to the noisy ones
        yy, xx = np.mgrid[:H, :W]
        masks[0] = (yy - H / 2) ** 2 + (xx - W / 3) ** 2 < (H / 3) ** 2
    cm = torch.from_numpy(np.ascontiguousarray(masks.transpose(0, 2, 1)).astype(np.uint8)).cuda().view(n, -1)   # column-major
    counts, n_runs = ops.rle_encode(cm)
    counts, n_runs = counts.cpu().numpy(), n_runs.cpu().numpy()
    for i in range(n):
        ref = rle_ref.rle_encode(masks[i].tolist()) if H * W <= 20000 else rle_ref.rle_encode_np(masks[i])   # the checker at every size (loop /
        assert n_runs[i] == len(ref)                                                                          # vectorised: tests/test_oracle_rle.py)
        got = counts[i, :n_runs[i]].tolist()
        assert got == ref
        assert (rle.counts_to_mask(got, H, W) == masks[i]).all()
        assert rle.counts_to_string(got).decode("ascii") == rle_ref.rle_to_string(got)
        assert rle_ref.rle_from_string(rle_ref.rle_to_string(got)) == got


def test_rle_buffer_overflow_is_reported():
    from openvis_amd import ops
    m = (torch.arange(4096) % 2).to(torch.uint8).cuda().view(1, -1)          # 4096 runs (+ leading zero-length... starts with 0)
    counts, n_runs = ops.rle_encode(m, cap=100)
    assert int(n_runs[0]) == 4096 and int(n_runs[0]) > 100


def test_model_rle_output_equals_dense_masks():
    import bench
    from openvis_amd import config, weights, rle
    from openvis_amd.catalog import MetadataCatalog
    from tests.test_openvis_gpu import CLIP_ARCH, K, _frames
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    sd = weights.random_init(weights.openvis_spec("r50", CLIP_ARCH, 100), seed=7)
    outs = []
    for flag in (False, True):
        cfg = config.get_cfg()
        cfg.MODEL.MASK_FORMER.TEST.OUTPUT_RLE = flag
        model = config.build_model(cfg)
        model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision="fp16")
        model.load_state_dict(sd)
        names = [f"class_{i}" for i in range(K)]
        MetadataCatalog.get("synthetic_val").set(thing_classes=names)
        model.clip_adapter.set_text_features(names, bench.synth_text(K, CLIP_ARCH["embed_dim"]))
        outs.append(model([{"image": [f for f in _frames()], "dataset_name": "synthetic_val", "height": 97, "width": 131}]))
    dense, enc = outs
    assert "pred_masks" not in enc and len(enc["pred_masks_rle"]) == len(dense["pred_masks"]) == 10
    assert dense["pred_labels"] == enc["pred_labels"]
    for j in range(10):
        for t in range(dense["pred_masks"][j].shape[0]):
            r = enc["pred_masks_rle"][j][t]
            assert r["size"] == [97, 131]
            m = rle.counts_to_mask(rle.string_to_counts(r["counts"]), 97, 131)
            assert (m == dense["pred_masks"][j][t].numpy()).all()
