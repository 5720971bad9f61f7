"""GPU: window inference (minvis.py:340-362 / san.py:285-307 / openvis.py:283-305) of the online models: (1) against the REFERENCE's
own `run_window_inference` methods (tests/golden/window_inference.npz, oracle/make_golden.py `window`), (2) equal to the
un-windowed run: per-frame stages are independent, windows only bound activation memory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arch,decoder", [("OpenVISOnline", "FrameMultiScaleMaskedTransformerDecoder"),
                                          ("SANOnline", "SideAdapterFrameMultiScaleMaskedTransformerDecoder"),
                                          ("BriVIS", "SideAdapterFrameMultiScaleMaskedTransformerDecoder")])
def test_window_inference_equals_full_clip(arch, decoder):
    import bench
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from openvis_amd.modeling.clip_adapter.side_adapter import SideAdapter
    from tests.test_openvis_gpu import CLIP_ARCH, K, _frames
    from tests.test_san_gpu import SAN_E2E_ARCH

    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_val").set(thing_classes=names)
    frames = torch.cat([_frames(s) for s in (0, 1, 2)])[:5]                 # 5 frames, windows of 2 -> 2 + 2 + 1
    outs, stages = [], []
    for window in (False, True):
        cfg = config.get_cfg()
        cfg.MODEL.META_ARCHITECTURE = arch
        cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = decoder
        cfg.MODEL.MASK_FORMER.TEST.WINDOW_INFERENCE = window
        cfg.MODEL.MASK_FORMER.TEST.WINDOW_SIZE = 2
        cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS = 4
        cfg.MODEL.PRECISION = "fp32"
        model = config.build_model(cfg)
        if arch == "OpenVISOnline":
            sd = weights.random_init(weights.openvis_spec("r50", CLIP_ARCH, 100), seed=9)
            model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision="fp32")
            dim = CLIP_ARCH["embed_dim"]
        else:
            sd = weights.random_init(weights.brivis_spec("r50", SAN_E2E_ARCH, 100), seed=9)
            model.clip_adapter = SideAdapter("tiny", broken_idx=3, merge_ids=[1, 2, 3], num_queries=100, arch=SAN_E2E_ARCH, precision="fp32")
            dim = SAN_E2E_ARCH["embed_dim"]
        model.load_state_dict(sd)
        model.clip_adapter.set_text_features(names, bench.synth_text(K, dim))
        st = {}
        outs.append(model([{"image": [f for f in frames], "dataset_name": "synthetic_val"}], stages=st))
        stages.append(st)
    a, b = stages
    assert torch.equal(a["indices"].cpu(), b["indices"].cpu())
    pm_a, pm_b = a["pred_masks"].cpu(), b["pred_masks"].cpu()
    assert pm_a.shape == pm_b.shape and (pm_a - pm_b).abs().max().item() < 1e-3
    assert ((pm_a > 0) == (pm_b > 0)).float().mean().item() > 0.9999
    assert (a["probs"].cpu() - b["probs"].cpu()).abs().max().item() < 1e-4
    assert outs[0]["pred_labels"] == outs[1]["pred_labels"]
    for m0, m1 in zip(outs[0]["pred_masks"], outs[1]["pred_masks"]):
        assert (m0 != m1).float().mean().item() < 1e-4


@pytest.mark.parametrize("arch", ["OpenVISOnline", "SANOnline"])
def test_window_inference_matches_the_reference_golden(arch):
    """The HIP path's backbone -> sem_seg_head windows (MODEL.MASK_FORMER.TEST.WINDOW_SIZE 2 on 5 frames: 2 + 2 + 1) and the tracker
    against the outputs of the reference's MinVIS.run_window_inference + post_processing / SANOnline.run_window_inference."""
    from openvis_amd import config
    from tests.test_oracle_path import load_window_case

    g, Wd, Ws, frames, mg, win = load_window_case()
    san = arch == "SANOnline"
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = arch
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = ("SideAdapterFrameMultiScaleMaskedTransformerDecoder" if san
                                                      else "FrameMultiScaleMaskedTransformerDecoder")
    cfg.MODEL.MASK_FORMER.TEST.WINDOW_INFERENCE = True
    cfg.MODEL.MASK_FORMER.TEST.WINDOW_SIZE = win
    cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS = 4
    cfg.MODEL.PRECISION = "fp32"
    model = config.build_model(cfg)
    sd = Ws if san else Wd
    model.backbone.load_state_dict(sd, "backbone.", model.device)                      # (no CLIP tower: the rows under test end at the tracker)
    model.sem_seg_head.load_state_dict(sd, "sem_seg_head.", model.device)
    assert model.window_inference and model.window_size == win
    images, _, _ = model.preprocess(frames.to(model.device))
    mg_d = [x.permute(0, 2, 3, 1).contiguous().to(model.device) for x in mg]            # channel-last, like the side adapter's features
    calls = []

    def per_window(b0, b1):
        calls.append((b0, b1))
        extra = [x[b0:b1] for x in mg_d] if san else None
        return model.sem_seg_head(model.backbone(images[b0:b1]), extra_feats=extra)

    run = model.run_window_inference if san else (lambda fn, T: model._windowed(model, fn, T))
    out = run(per_window, images.shape[0])
    assert calls == [(0, 2), (2, 4), (4, 5)]
    torch.cuda.synchronize()
    emb = out["pred_embeds"].cpu().numpy()
    if san:
        assert np.abs(emb - g["san_pred_embeds"]).max() < 2e-3
        pm, ref = out["pred_masks"].cpu().numpy(), g["san_pred_masks"]
        assert np.abs(pm - ref).max() < 5e-3 and ((pm > 0) == (ref > 0)).mean() > 0.9995
        assert np.abs(out["class_attn_biases"].cpu().numpy() - g["san_class_attn_biases"]).max() < 5e-3
        from openvis_amd.modeling.minvis import batch_video_match_via_embeds
        idx, _ = batch_video_match_via_embeds(torch.from_numpy(g["san_pred_embeds"]).to(model.device))
        assert np.array_equal(idx.cpu().numpy(), g["san_indices"])                         # tracker on the reference's embeddings: exact
        return
    assert np.abs(emb - g["pred_embeds"]).max() < 2e-3
    # the reference's outputs before the tracker = its tracked outputs with the permutation undone: post[t, q] = pre[t, indices[t, q]]
    idx_ref = g["indices"][0]                                                              # [T, Q]
    T, Q = idx_ref.shape
    pre_masks = np.empty_like(g["post_pred_masks"])                                        # [1, Q, T, h, w]
    pre_logits = np.empty_like(g["post_pred_logits"])                                      # [1, T, Q, C]
    for t in range(T):
        pre_masks[0, idx_ref[t], t] = g["post_pred_masks"][0, :, t]
        pre_logits[0, t, idx_ref[t]] = g["post_pred_logits"][0, t]
    pm = out["pred_masks"].cpu().numpy()
    assert np.abs(pm - pre_masks).max() < 5e-3 and ((pm > 0) == (pre_masks > 0)).mean() > 0.9995
    assert np.abs(out["pred_logits"].cpu().numpy() - pre_logits).max() < 2e-3
    # the tracker (minvis.py:320-338) on the REFERENCE's embeddings must give the reference's assignment exactly; on the HIP run's own
    # embeddings (2e-3 away, random-init queries are near-duplicates) the assignment may legitimately differ in near-tied pairs, so
    # the tracked tensors are checked through the permutation the HIP run itself produced
    from openvis_amd.modeling.minvis import batch_video_match_via_embeds
    idx_on_ref, _ = batch_video_match_via_embeds(torch.from_numpy(g["pred_embeds"]).to(model.device))
    assert np.array_equal(idx_on_ref.cpu().numpy(), g["indices"])
    post = model._post(model, out)
    idx = post["indices"].cpu().numpy()[0]
    print("window golden: tracker assignment on the HIP embeddings equals the reference's in %d of %d (frame, query) slots" % ((idx == idx_ref).sum(), idx.size))
    assert (idx == idx_ref).mean() > 0.9
    ppm = post["pred_masks"].cpu().numpy()
    for t in range(T):
        assert sorted(idx[t].tolist()) == list(range(Q))                                   # a permutation
        assert np.array_equal(ppm[0, :, t], pm[0, idx[t], t])                              # batch_index: pure data movement
