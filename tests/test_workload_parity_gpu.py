"""GPU: parity at WORKLOAD size for BASELINE.json configs[2], [3], [4] under the precision policy
bench.py times (config defaults), against golden vectors of the f32 CPU oracle (oracle/make_golden_workload.py; inputs are
re-generated from seeds, only the oracle's outputs are stored):

  C3  SANOnline R50 + SideAdapter ViT-B/16, 5 frames of 720x1280                     (reference: openvis/san.py:177-283)
  C4  BriVIS R50, one 36-frame 720p clip: Hungarian linker over all 36 frames, resampler, heads   (brivis.py:131-176,
      resampler.py:244-316); masks compared on frames 0 / 17 / 35, pixel counts on all 36
  C5  BriVIS Swin-L + SideAdapter ViT-L/14@336, 1 frame of 1080x1920                  (swin/brivis_SwinB_*.yaml:5-22 + Swin-L block)

North star: instance ids identical, every query mask IoU >= 0.999, cosine logits within 1e-3.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
K, Q = 40, 100


def _build(arch, backbone="r50", clip="ViT-B/16", split="auto"):
    import bench
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import _CLIP_ARCH
    cfg = config.get_cfg()                                   # defaults = what bench.py --model san_online / brivis runs
    cfg.MODEL.META_ARCHITECTURE = arch
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "SideAdapterFrameMultiScaleMaskedTransformerDecoder"
    cfg.MODEL.CLIP_ADAPTER.CLIP_MODEL_NAME = clip
    if clip == "ViT-L/14@336px":
        cfg.MODEL.CLIP_ADAPTER.MERGE_IDS, cfg.MODEL.CLIP_ADAPTER.BROKEN_ID = [6, 12, 18], 21
        cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS, cfg.MODEL.CLIP_ADAPTER.CLIP_EMBED_DIMS = 16, 768
    if backbone != "r50":
        a = weights.SWIN_ARCH[backbone]
        cfg.MODEL.BACKBONE.NAME = "D2SwinTransformer"
        cfg.MODEL.SWIN.EMBED_DIM, cfg.MODEL.SWIN.DEPTHS = a["embed_dim"], list(a["depths"])
        cfg.MODEL.SWIN.NUM_HEADS, cfg.MODEL.SWIN.WINDOW_SIZE = list(a["num_heads"]), a["window"]
    assert cfg.MODEL.PRECISION == "mixed"
    cfg.MODEL.F32_GEMM_SPLIT = split
    model = config.build_model(cfg)
    assert model.f32_gemm_mode == config.F32_GEMM_SPLITS["fp16x2" if split == "auto" else split]
    spec = weights.san_spec(backbone, _CLIP_ARCH[clip], Q) if arch == "SANOnline" else weights.brivis_spec(backbone, _CLIP_ARCH[clip], Q)
    sd = weights.random_init(spec, seed=42)
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_workload").set(thing_classes=names)
    model.clip_adapter.set_text_features(names, bench.synth_text(K, _CLIP_ARCH[clip]["embed_dim"]))
    scale = float(sd["clip_adapter.clip_model.logit_scale"].exp())
    return model, scale


def _per_query_iou(got_logits, ref_bits, ref_shape):
    """got_logits [Q,T,h,w] (device/cpu tensor) vs packed sign bits of the oracle's logits."""
    ref = np.unpackbits(ref_bits, axis=-1)[..., : int(ref_shape[-1])].astype(bool)
    g = (got_logits.cpu() > 0).numpy()
    assert g.shape == tuple(int(x) for x in ref_shape) == ref.shape
    inter = (g & ref).sum(axis=(1, 2, 3)).astype(np.float64)
    union = (g | ref).sum(axis=(1, 2, 3)).astype(np.float64)
    return np.where(union > 0, inter / np.maximum(union, 1), 1.0)


def _check_topk(out, gold, tol=2e-3):
    sg = {(int(q), int(l)): s for q, l, s in zip(out["pred_queries"], out["pred_labels"], out["pred_scores"])}
    sr = {(int(q), int(l)): float(s) for q, l, s in zip(gold["top_rows"], gold["top_labels"], gold["top_scores"])}
    both = set(sg) & set(sr)
    assert len(both) >= 9, (sorted(sg), sorted(sr))         # topk(sorted=False) on near-ties may swap the 10th entry
    assert max(abs(sg[k] - sr[k]) for k in both) < tol


def _assert_bit_exact_outside_ambiguous(tag, got_logits, g, split):
    """every mask bit equals the oracle's except where |oracle logit| < 1e-3.  Measured (MI355X, round 5): C3 26-30 of 29.4 M bits differ,
    2-3 of them at 1e-4 <= |logit| < 1e-3; C4 30-60 of 17.7 M (0-4 beyond 1e-4; bf16x2: 337, 110 beyond 1e-4); C5 21 of 26.1 M, all below 1e-4;
    none beyond 1e-3 under any split."""
    from tests._logits import differing_bits_outside_ambiguous
    ref = np.unpackbits(g["mask_bits"], axis=-1)[..., : int(g["mask_shape"][-1])].astype(bool)
    n_diff, outside = differing_bits_outside_ambiguous((got_logits.cpu() > 0).numpy(), ref, g)
    print("%s [%s]: %d of %d mask bits differ; outside the |oracle logit| < eps sets: %s" % (tag, split, n_diff, ref.size, outside))
    assert outside[1e-3] == 0, outside


@pytest.mark.parametrize("split", ["auto", "bf16x3"])
def test_c3_san_online_720p_under_the_bench_policy(split):
    """configs[2] at its full T = 5 frames"""
    import bench
    g = np.load(os.path.join(GOLDEN, "c3_san_online_720p.npz"))
    model, scale = _build("SANOnline", split=split)
    assert (model.backbone.precision, model.clip_adapter.precision) == ("fp32", "fp32")      # the SAN-family "auto" policy
    T3 = int(g["mask_shape"][1])
    assert T3 == 5
    frames = bench.synth_frames(T3, 720, 1280, 3, "cpu")
    st = {}
    out = model([{"image": [f for f in frames], "dataset_name": "synthetic_workload"}], stages=st)
    torch.cuda.synchronize()
    idx = st["indices"].cpu().numpy().reshape(g["indices"].shape)
    assert np.array_equal(idx, g["indices"])                                                   # instance ids identical
    iou = _per_query_iou(st["pred_masks"][0], g["mask_bits"], g["mask_shape"])
    print("C3 per-query IoU: min %.5f median %.5f" % (iou.min(), np.median(iou)))
    assert iou.min() >= 0.999, (iou.min(), int((iou < 0.999).sum()))
    _assert_bit_exact_outside_ambiguous("C3", st["pred_masks"][0], g, split)
    dl = np.abs(st["pred_logits"][0].cpu().numpy() - g["logits"]).max() / scale
    print("C3 max |cos diff| %.2e" % dl)
    assert dl <= 1e-3
    assert np.abs(st["probs"].cpu().numpy() - g["probs"]).max() < 1e-3
    _check_topk(out, g)
    assert len(out["pred_masks"]) == 10 and tuple(out["pred_masks"][0].shape) == (T3, 720, 1280)


def _brivis_case(gold_name, backbone, clip, T, H, W, seed, split="auto"):
    import bench
    g = np.load(os.path.join(GOLDEN, gold_name))
    model, scale = _build("BriVIS", backbone, clip, split)
    assert (model.backbone.precision, model.resampler.precision) == ("fp32", "fp32")
    frames = bench.synth_frames(T, H, W, seed, "cpu")
    st = {}
    out = model([{"image": [f for f in frames], "dataset_name": "synthetic_workload"}], stages=st)
    torch.cuda.synchronize()
    idx = st["indices"].cpu().numpy().reshape(g["indices"].shape)
    assert np.array_equal(idx, g["indices"]), int((idx != g["indices"]).sum())                # ids identical on ALL frames
    keep = [int(t) for t in g["keep_frames"]]
    pm = st["pred_masks"][0]                                                                   # [Q,T,h,w]
    iou = _per_query_iou(pm[:, keep], g["mask_bits"], g["mask_shape"])
    ref_bits = np.unpackbits(g["mask_bits"], axis=-1)[..., : int(g["mask_shape"][-1])].astype(bool)
    got_bits = (pm[:, keep].cpu() > 0).numpy()
    print("%s [%s] per-query IoU on frames %s: min %.5f median %.5f; exact bit match rate %.6f (%d of %d bits differ), %d of %d query masks "
          "bit-identical" % (gold_name, split, keep, iou.min(), np.median(iou), (got_bits == ref_bits).mean(), int((got_bits != ref_bits).sum()),
                             got_bits.size, int((got_bits == ref_bits).all(axis=(1, 2, 3)).sum()), got_bits.shape[0]))
    assert model.f32_gemm_mode == __import__("openvis_amd.config", fromlist=["x"]).F32_GEMM_SPLITS["fp16x2" if split == "auto" else split]  # no fall-back
    assert iou.min() >= 0.999, (iou.min(), int((iou < 0.999).sum()))
    _assert_bit_exact_outside_ambiguous(gold_name, pm[:, keep], g, split)
    # every frame: positive-pixel count of every (query, frame) mask within 0.2 % of the mask area of the oracle's
    cnt = (pm > 0).sum(dim=(-1, -2)).cpu().numpy()
    area = pm.shape[-1] * pm.shape[-2]
    assert np.abs(cnt - g["mask_counts"]).max() <= 0.002 * area, np.abs(cnt - g["mask_counts"]).max()
    lg = st["pred_logits"][0].cpu().numpy()                                                    # [T,Q,K+1]
    dl = np.abs(lg[keep] - g["logits_subset"]).max() / scale
    print("%s max |cos diff| %.2e" % (gold_name, dl))
    assert dl <= 1e-3
    assert np.abs(st["probs"].cpu().numpy() - g["probs"]).max() < 1e-3
    _check_topk(out, g)
    assert len(out["pred_masks"]) == 10 and tuple(out["pred_masks"][0].shape) == (T, H, W)


@pytest.mark.parametrize("split", ["auto", "bf16x3", "bf16x2"])
def test_c4_brivis_36_frames_720p_under_the_bench_policy(split):
    """auto = fp16x2 (what bench.py times); the f32-grade bf16x3 and the 16-bit bf16x2 beside it, exact-bit rates printed"""
    _brivis_case("c4_brivis_720p_36f.npz", "r50", "ViT-B/16", 36, 720, 1280, 1000, split)


def test_c5_brivis_swinl_vitl14_336_1080p_under_the_bench_policy():
    _brivis_case("c5_brivis_swinl_1080p.npz", "swin_l", "ViT-L/14@336px", 3, 1080, 1920, 1000)


def test_c5_brivis_swinl_vitl14_336_1080p_at_the_full_36_frames():
    """configs[4] at its full T = 36 frames of 1080x1920 under an assertion (round-4 review: it had only run in a scratch bench): identical
    track ids on all 36 frames, every query mask IoU >= 0.999 and bit-exact outside |oracle logit| < 1e-3 on frames 0 / 17 / 35, pixel counts
    of every (frame, query) mask within 0.2 % of the mask area, cosine logits within 1e-3."""
    _brivis_case("c5_brivis_swinl_1080p_36f.npz", "swin_l", "ViT-L/14@336px", 36, 1080, 1920, 1000)
