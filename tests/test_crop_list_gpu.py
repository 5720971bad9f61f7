"""GPU: MODEL.CLIP_ADAPTER.CROP_LIST -- the crop list of ClipAdapter (adapter.py:86-102: boxes of the binary masks, valid flags, one crop per
valid (frame, query)) built on the device without the reference's host read-back, against the host (compacting) path: same class
probabilities for every query that has a crop, same top-10, same output masks -- with empty masks mixed in, with none, and with only empty
masks (openvis.py:127-128: the empty result)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(crop_list):
    import bench
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from tests.test_openvis_gpu import CLIP_ARCH
    names = [f"class_{i}" for i in range(7)]
    MetadataCatalog.get("synthetic_croplist").set(thing_classes=names)
    model = config.build_model(config.get_cfg())
    model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision="fp16")
    model.clip_adapter.crop_list = crop_list
    model.load_state_dict(weights.random_init(weights.openvis_spec("r50", CLIP_ARCH, 100), seed=3))
    model.clip_adapter.set_text_features(names, bench.synth_text(7, CLIP_ARCH["embed_dim"], spread=0.25))
    return model, names


def _masks(kind, g):
    m = torch.randn(100, 3, 24, 32, generator=g) * 2
    if kind == "mixed":                                        # a third of the (query, frame) masks empty, four queries empty on every frame
        empty = torch.rand(100, 3, generator=g) < 0.33
        empty[[5, 17, 63, 99]] = True
        m[empty] = -5.0 - m[empty].abs()
    elif kind == "none":
        m = -5.0 - m.abs()
    return m.cuda()


@pytest.mark.parametrize("kind", ["mixed", "full", "none"])
def test_device_crop_list_equals_host_crop_list(kind):
    import bench
    from openvis_amd.modeling.clip_adapter.adapter import DeviceCrops
    g = torch.Generator().manual_seed(2)
    masks = _masks(kind, g)
    frames = bench.synth_frames(3, 96, 128, 4, "cuda")
    outs = {}
    for mode in ("host", "device"):
        model, names = _model(mode)
        probs, row_ids, extras = model.open_vocabulary_inference(torch.zeros(100, 2, device="cuda"), masks, frames, names, (96, 128))
        if mode == "device":
            dc = extras["device_crops"]
            assert isinstance(dc, DeviceCrops) and tuple(dc.crops.shape) == (300, 6) and tuple(dc.slot.shape) == (3, 100)
            extras = model._host_view_of_crops(extras)
            assert int(dc.counts.item()) == int(extras["valid"].sum())
        out = model.inference_video(100, 7, probs, row_ids, masks, (96, 128), (96, 128), 96, 128,
                                    n_valid=extras["device_crops"].counts if mode == "device" else None)
        outs[mode] = (probs, row_ids, extras, dict(out.items()) if hasattr(out, "items") else out)
    (ph, rh, eh, oh), (pd, rd, ed, od) = outs["host"], outs["device"]
    if kind == "none":
        assert ph is None and not eh["valid"].any() and not ed["valid"].any()
        assert oh["pred_masks"] == [] and od["pred_masks"] == [] and od["pred_scores"] == [] and od["pred_labels"] == []
        return
    assert np.array_equal(eh["valid"], ed["valid"]) and np.array_equal(np.asarray(eh["crops"]), ed["crops"])
    assert torch.equal(eh["crop_logits"], ed["crop_logits"])                                    # same crops -> the same tower rows, bit for bit
    qv = torch.from_numpy(eh["valid"].any(axis=0)).cuda()
    assert torch.equal(ph[qv], pd[qv]) and (pd[~qv] == -1).all()                               # rows without a crop can never win the top-k
    assert sorted(zip(oh["pred_queries"], oh["pred_labels"])) == sorted(zip(od["pred_queries"], od["pred_labels"]))
    so, sd_ = dict(zip(zip(oh["pred_queries"], oh["pred_labels"]), oh["pred_scores"])), dict(zip(zip(od["pred_queries"], od["pred_labels"]), od["pred_scores"]))
    assert so == sd_
    mo = {k: m for k, m in zip(zip(oh["pred_queries"], oh["pred_labels"]), oh["pred_masks"])}
    md = {k: m for k, m in zip(zip(od["pred_queries"], od["pred_labels"]), od["pred_masks"])}
    assert all(torch.equal(mo[k], md[k]) for k in mo)


def test_auto_mode_switches_to_the_device_list_after_a_clip_with_mostly_valid_masks():
    """auto: the first clip reads the boxes back (and learns the share of non-empty masks); from then on the list is built on the device
    while that share stays >= 90 % -- the count of a device-list clip rides back with its outputs and decides two clips later."""
    import bench
    model, names = _model("auto")
    ad = model.clip_adapter
    frames = bench.synth_frames(2, 96, 128, 7, "cuda")
    inp = [{"image": [f for f in frames], "dataset_name": "synthetic_croplist"}]
    st = {}
    o1 = model(inp, stages=st)
    o1.wait()
    assert "device_crops" not in st and ad._valid_frac is not None                            # host path, share now known
    frac = ad._valid_frac
    st2 = {}
    o2 = model(inp, stages=st2)
    o2.wait()
    assert ("device_crops" in st2) == (frac >= 0.9)
    assert o1["pred_labels"] == o2["pred_labels"] and o1["pred_scores"] == o2["pred_scores"]
    assert all(torch.equal(a, b) for a, b in zip(o1["pred_masks"], o2["pred_masks"]))
    if frac >= 0.9:
        # the count of a device-list clip is consulted TWO clips later, deterministically (never "whatever has arrived by now")
        assert len(ad._pending_counts) == 1
        ad._valid_frac = -1.0
        assert ad._use_device_list() is False and len(ad._pending_counts) == 1                 # clip n - 1's count is not looked at
        ad._valid_frac = frac
        o3 = model(inp)
        o3.wait()
        assert len(ad._pending_counts) == 2
        assert ad._use_device_list() and len(ad._pending_counts) == 1 and abs(ad._valid_frac - frac) < 1e-6


def test_device_list_raises_when_fewer_pairs_than_topk_exist():
    """Device crop list: every query row takes part in the top-k and rows without a crop hold -1.  With fewer (valid query, class) pairs
    than topk the reference's `topk(10)` raises (video_maskformer.py:269) and so does the host-list path; the device-list path must not
    hand out the -1 filler rows as detections (round-4 review)."""
    model, names = _model("device")
    Q, K, T, h, w = 100, 3, 2, 24, 32
    probs = torch.full((Q, K), -1.0, device="cuda")
    probs[[5, 17]] = torch.tensor([[0.2, 0.3, 0.5], [0.6, 0.3, 0.1]], device="cuda")      # 2 valid queries x 3 classes = 6 pairs < 10
    rid = torch.arange(Q, dtype=torch.int32, device="cuda")
    masks = torch.randn(Q, T, h, w, device="cuda")
    nv = torch.tensor([3], dtype=torch.int32, device="cuda")
    out = model.inference_video(Q, K, probs, rid, masks, (4 * h, 4 * w), (90, 120), 90, 120, n_valid=nv)
    with pytest.raises(RuntimeError, match="out of range"):
        out.wait() if hasattr(out, "wait") else None
        out["pred_scores"]
    # ... and with at least topk pairs the same call returns them
    probs[[1, 2]] = torch.tensor([[0.1, 0.2, 0.7], [0.3, 0.3, 0.4]], device="cuda")       # 4 x 3 = 12 pairs
    out = model.inference_video(Q, K, probs, rid, masks, (4 * h, 4 * w), (90, 120), 90, 120, n_valid=nv)
    assert len(out["pred_scores"]) == 10 and min(out["pred_scores"]) > 0 and set(out["pred_queries"]) <= {1, 2, 5, 17}
