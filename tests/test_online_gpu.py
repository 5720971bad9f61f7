"""GPU: per-frame decoder, Hungarian linker (A14) and the OpenVISOnline meta-architecture vs the oracle / reference goldens."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from tests._synth import synth_weights, synth_inputs

pytestmark = pytest.mark.gpu


def _spec(arr):
    return [(k, tuple(s)) for k, s in json.loads(bytes(arr.tolist()).decode())]


def test_hungarian_link_matches_reference_tracker():
    from openvis_amd import ops
    g = np.load(os.path.join(GOLDEN, "frame_decoder_tracker.npz"))
    idx = ops.hungarian_link(torch.from_numpy(g["track_seq"]).cuda()).cpu().numpy()
    assert np.array_equal(idx, g["track_indices"][0])                    # reference: scipy chain, minvis.py:44-72
    idx = ops.hungarian_link(torch.from_numpy(g["pred_embeds"][0]).cuda()).cpu().numpy()
    assert np.array_equal(idx, g["indices"][0])


def test_hungarian_link_random_vs_scipy():
    from openvis_amd import ops
    from oracle import torch_ref as TR
    gen = torch.Generator().manual_seed(5)
    for (T, Q, C) in ((4, 100, 256), (3, 17, 32), (2, 130, 64)):
        emb = torch.randn(T, Q, C, generator=gen)
        ref, ref_emb = TR.video_match_via_embeds(emb)
        idx = ops.hungarian_link(emb.cuda())
        assert np.array_equal(idx.cpu().numpy(), ref.numpy())
        out = torch.empty_like(emb).cuda()
        ops.batch_index_rows(emb.cuda(), idx, out, Q * C, C, Q * C, C, C)
        assert torch.equal(out.cpu(), ref_emb)


def test_frame_decoder_matches_reference_golden():
    from openvis_amd.modeling.transformer_decoder import FrameMultiScaleMaskedTransformerDecoder
    g = np.load(os.path.join(GOLDEN, "frame_decoder_tracker.npz"))
    s_dec, s_ms, s_mf, _ = [int(x) for x in g["seeds"]]
    T = g["pred_masks"].shape[2]
    sd = synth_weights(_spec(g["spec"]), s_dec, "sem_seg_head.predictor.")
    dec = FrameMultiScaleMaskedTransformerDecoder(256, True, num_classes=1, hidden_dim=256, num_queries=100, nheads=8,
                                                  dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256,
                                                  enforce_input_project=False, num_frames=T, precision="fp32")
    dec.load_state_dict(sd, "sem_seg_head.predictor.", "cuda")
    ms = synth_inputs([(T, 256, 2, 3), (T, 256, 4, 6), (T, 256, 8, 12)], s_ms)
    mf = synth_inputs([(T, 256, 16, 24)], s_mf)[0]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    out = dec([nhwc(m) for m in ms], nhwc(mf))
    pm = out["pred_masks"].cpu().numpy()
    assert np.abs(pm - g["pred_masks"]).max() < 2e-3
    assert ((pm > 0) == (g["pred_masks"] > 0)).mean() > 0.9999
    assert np.abs(out["pred_embeds"].cpu().numpy() - g["pred_embeds"]).max() < 2e-4
    assert np.abs(out["pred_logits"].cpu().numpy() - g["pred_logits"]).max() < 2e-4


CLIP_ARCH = dict(width=256, layers=2, heads=4, patch=16, resolution=64, embed_dim=64)


@pytest.mark.parametrize("policy", ["fp32", "mixed"])
def test_openvis_online_end_to_end(policy):
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from oracle import torch_ref as TR
    from tests.test_openvis_gpu import _frames, K, H, W

    spec = weights.resnet50_spec() + weights.pixel_decoder_spec() + weights.video_decoder_spec() + weights.clip_visual_spec(**CLIP_ARCH)
    sd = weights.random_init(spec, seed=11)
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "OpenVISOnline"
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "FrameMultiScaleMaskedTransformerDecoder"
    cfg.MODEL.PRECISION = policy
    model = config.build_model(cfg)
    model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision="fp32" if policy == "fp32" else "fp16")
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_val").set(thing_classes=names)
    gen = torch.Generator().manual_seed(1)
    base = torch.randn(1, CLIP_ARCH["embed_dim"], generator=gen)
    text = torch.nn.functional.normalize(base + 0.05 * torch.randn(K, CLIP_ARCH["embed_dim"], generator=gen), dim=-1)
    model.clip_adapter.set_text_features(names, text)
    frames = _frames(3)
    st = {}
    out = model([{"image": [f for f in frames], "dataset_name": "synthetic_val"}], stages=st)
    ref_st = {}
    with torch.no_grad():
        ref = TR.openvis_online_forward(frames, sd, text, stages=ref_st, clip_heads=CLIP_ARCH["heads"],
                                        clip_resolution=CLIP_ARCH["resolution"])
    # tracker assignment (instance ids) identical
    assert np.array_equal(st["indices"].cpu().numpy()[0], ref_st["indices"].numpy())
    g, r = st["pred_masks"].cpu(), ref_st["pred_masks"]
    inter, union = ((g > 0) & (r > 0)).sum().item(), ((g > 0) | (r > 0)).sum().item()
    assert inter / max(union, 1) > 0.999, inter / max(union, 1)
    assert out["image_size"] == ref["image_size"] == (H, W)
    rows_ref = ref_st["valid"].any(0).nonzero()[:, 0].tolist()
    sg = {(q, l): s for q, l, s in zip(out["pred_queries"], out["pred_labels"], out["pred_scores"])}
    sr = {(rows_ref[rr], l): s for rr, l, s in zip(ref["rows"], ref["pred_labels"], ref["pred_scores"])}
    common = set(sg) & set(sr)
    assert len(common) >= 8
    tol = 2e-3 if policy == "fp32" else 3e-2
    assert max(abs(sg[k] - sr[k]) for k in common) < tol


@pytest.mark.parametrize("name", ["embedding_frame", "proposal_frame", "embedding_video", "proposal_video"])
def test_decoder_head_variants_match_reference_golden(name):
    """Embedding* / Proposal* decoders (frame decoder:157-207, video decoder:487-537; configs/openvoc_ytvis_coco/simplebsl*.yaml): the parent
    decoder with another class head, against outputs of the reference's own classes (oracle/make_golden.py heads).  The variants carry the
    parent fixtures' weights and inputs, so their masks must also match those fixtures."""
    from openvis_amd.modeling import transformer_decoder as TD
    from openvis_amd.modeling.transformer_decoder import frame_mask2former_transformer_decoder as FD, video_mask2former_transformer_decoder as VD
    g = np.load(os.path.join(GOLDEN, "decoder_head_variants.npz"))
    gf = np.load(os.path.join(GOLDEN, "frame_decoder_tracker.npz"))
    gv = np.load(os.path.join(GOLDEN, "pixel_decoder_decoder.npz"))
    frame = name.endswith("frame")
    cls = {"embedding_frame": FD.EmbeddingFrameMultiScaleMaskedTransformerDecoder, "proposal_frame": FD.ProposalFrameMultiScaleMaskedTransformerDecoder,
           "embedding_video": VD.EmbeddingVideoMultiScaleMaskedTransformerDecoder, "proposal_video": VD.ProposalVideoMultiScaleMaskedTransformerDecoder}[name]
    kw = dict(in_channels=256, num_classes=1, hidden_dim=256, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9, pre_norm=False,
              mask_dim=256, enforce_input_project=False, precision="fp32")
    if frame:
        s_dec, s_ms, s_mf, _ = [int(x) for x in gf["seeds"]]
        T, parent_spec, ref_masks = 3, _spec(gf["spec"]), gf["pred_masks"]
        ms = synth_inputs([(T, 256, 2, 3), (T, 256, 4, 6), (T, 256, 8, 12)], s_ms)
        mf = synth_inputs([(T, 256, 16, 24)], s_mf)[0]
    else:
        s_dec, T, parent_spec, ref_masks = int(gv["seeds"][2]), 2, _spec(gv["spec_dec"]), gv["pred_masks"]
        ms = [torch.from_numpy(gv[f"ms{i}"]) for i in range(3)]
        mf = torch.from_numpy(gv["mask_features"])
    extra = dict(clip_dims=int(g["clip_dims"][0])) if name.startswith("embedding") else {}
    dec = cls(mask_classification=True, num_frames=T, **extra, **kw)
    pre = "sem_seg_head.predictor."
    sd = {k: v for k, v in synth_weights(parent_spec, s_dec, pre).items() if not k.startswith(pre + "class_embed")}
    sd.update(synth_weights(_spec(g[name + "_head_spec"]), int(g["head_seed"][0]), pre))
    dec.load_state_dict(sd, pre, "cuda")
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    out = dec([nhwc(m) for m in ms], nhwc(mf))
    pm = out["pred_masks"].cpu().numpy()
    assert np.abs(pm - ref_masks).max() < 2e-3 and ((pm > 0) == (ref_masks > 0)).mean() > 0.9999
    lg = out["pred_logits"].cpu().numpy()
    assert lg.shape == g[name + "_logits"].shape
    assert np.abs(lg - g[name + "_logits"]).max() < 2e-4 * max(1.0, float(np.abs(g[name + "_logits"]).max()))
