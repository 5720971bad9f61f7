"""CPU: the META-ARCHITECTURE GLUE of the oracle (oracle/torch_ref.py) against fixtures produced by the REFERENCE's own functions
(oracle/make_golden_glue.py -> tests/golden/glue_functions.npz, glue_forward.npz): OpenVIS.open_vocabulary_inference,
VideoMaskFormer.postprocess / inference_video, ClipAdapter._preprocess_image / encode_image, MinVIS.post_processing,
BriVIS.reset_image_output_order / post_processing, batch_index -- alone on crafted inputs, and inside the reference's whole eval
`forward` of OpenVIS (with ClipAdapter and with AdaptedClipAdapter) / OpenVISOnline / SAN / SANOnline / BriVIS.  The end-to-end config tests (C1-C5) compare the HIP path with these
oracle functions; this file is what ties them to the reference."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as TR
from tests._synth import synth_weights

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _spec(arr):
    return [(k, tuple(s)) for k, s in json.loads(bytes(arr.tolist()).decode())]


def _bits(packed, shape):
    n = int(np.prod(shape))
    return np.unpackbits(packed)[:n].astype(bool).reshape(tuple(int(x) for x in shape))


@pytest.fixture(scope="module")
def gf():
    return np.load(os.path.join(GOLDEN, "glue_functions.npz"))


def test_aggregate_crop_logits_equals_open_vocabulary_inference(gf):
    """openvis.py:126-142 on recorded crop logits: 7 frames (chunks of 5 + 2, the second without a crop), two queries without a crop."""
    valid, logits, masks = (torch.from_numpy(gf[k]) for k in ("ovi_valid", "ovi_crop_logits", "ovi_masks"))
    probs, vmasks, _ = TR.aggregate_crop_logits(logits, valid, masks)
    assert np.array_equal(probs.numpy(), gf["ovi_probs"])
    assert np.array_equal(vmasks.numpy(), gf["ovi_masks_out"])
    assert TR.aggregate_crop_logits(logits[:0], torch.zeros_like(valid), masks)[:2] == ([], [])


def test_inference_video_equals_the_reference(gf):
    """video_maskformer.py:215-229, 262-298: x4 upsample, top-10 of the flattened scores, label = index % K, row = index // K,
    entropy of the selected rows, crop to the image size, resize to the output size, `> 0`."""
    low, cls = torch.from_numpy(gf["iv_lowres"]), torch.from_numpy(gf["iv_cls"])
    H, W, OH, OW = [int(x) for x in gf["iv_sizes"]]
    up = F.interpolate(low, size=(4 * low.shape[-2], 4 * low.shape[-1]), mode="bilinear", align_corners=False)
    for (oh, ow), key in (((OH, OW), "iv_masks"), ((H, W), "iv_masks_same")):
        vo = TR.inference_video(cls.shape[0], cls.shape[1], cls, up, (H, W), oh, ow)
        ref_masks = _bits(gf[key], gf[key + "_shape"])
        order = np.lexsort((gf["iv_labels"], gf["iv_scores"]))
        mine = np.lexsort((np.array(vo["pred_labels"]), np.array(vo["pred_scores"], np.float32)))
        assert np.array_equal(np.array(vo["pred_scores"], np.float32)[mine], gf["iv_scores"][order])
        assert np.array_equal(np.array(vo["pred_labels"])[mine], gf["iv_labels"][order])
        assert np.array_equal(np.array(vo["pred_entropys"], np.float32)[mine], gf["iv_entropys"][order])
        assert np.array_equal(torch.stack(vo["pred_masks"]).numpy()[mine], ref_masks[order])
        assert vo["image_size"] == (oh, ow)
    empty = TR.inference_video(7, 5, [], [], (H, W), OH, OW)
    assert empty["pred_scores"] == [] and empty["pred_masks"] == [] and empty["image_size"] == (OH, OW)
    # K + 1 columns: softmax, background column dropped (video_maskformer.py:218-219)
    bg = torch.from_numpy(gf["iv_cls_bg"])
    assert np.array_equal(F.softmax(bg, dim=-1)[:, :-1].numpy(), gf["iv_cls_bg_out"])


def test_clip_crops_equal_clipadapter_preprocess_image(gf):
    """adapter.py:73-116 + 140-142: valid flags, square boxes anchored top-left, roi_align of the UN-padded frame and of the padded
    soft mask, blend, /255, CLIP normalisation (BitMasks / roi_align themselves are un-vendored third-party on both sides)."""
    H, W, Hp, Wp, RES = [int(x) for x in gf["pp_sizes"]]
    frames = torch.from_numpy(gf["pp_frames"])
    low = torch.from_numpy(gf["pp_lowres"])
    up = F.interpolate(low, size=(Hp, Wp), mode="bilinear", align_corners=False)
    part_masks = up.sigmoid().transpose(0, 1).contiguous()
    regions, valid, boxes = TR.clip_crops(frames, part_masks, RES)
    assert np.array_equal(valid.numpy(), gf["pp_valid"])
    assert not valid[:, 3].any() and not valid[1, 5] and valid[0, 6]               # the crafted empty masks
    assert float(boxes[:, 2].max()) > W                                            # a square box reaching beyond the frame
    assert np.array_equal(regions.numpy(), gf["pp_regions"])
    assert np.array_equal(TR.clip_tower_input(regions, RES).numpy(), gf["pp_tower_input"])
    none = TR.clip_crops(frames, torch.zeros_like(part_masks), RES)
    assert none[0] is None and not none[1].any()


def test_tracker_post_processing_equals_the_reference(gf):
    """minvis.py:28-72, 320-338; brivis.py:231-265; utils/index.py:4-18."""
    embeds, logits, masks = (torch.from_numpy(gf[k]) for k in ("tr_embeds", "tr_logits", "tr_masks"))
    idx, fe = TR.video_match_via_embeds(embeds[0])
    assert np.array_equal(idx.numpy(), gf["tr_indices"][0])
    assert np.array_equal(fe.numpy(), gf["tr_frame_embeds"][0])
    post = TR.minvis_post_processing(dict(pred_logits=logits, pred_masks=masks, pred_embeds=embeds))
    assert np.array_equal(post["pred_logits"].numpy(), gf["tr_post_logits"])
    assert np.array_equal(post["pred_masks"].numpy(), gf["tr_post_masks"])
    # BriVIS.reset_image_output_order is the same re-ordering (brivis.py:231-240)
    assert np.array_equal(gf["tr_reset_logits"], gf["tr_post_logits"]) and np.array_equal(gf["tr_reset_masks"], gf["tr_post_masks"])
    K = logits.shape[-1] - 1
    cls, up = TR.temporal_mean_post_processing(logits, masks, (4 * masks.shape[-2], 4 * masks.shape[-1]), K)
    assert np.array_equal(cls.numpy(), gf["bv_cls"]) and np.array_equal(up.numpy(), gf["bv_masks"])
    # batch_index, both layouts
    src, bi = torch.from_numpy(gf["bi_src"]), torch.from_numpy(gf["bi_idx"])
    assert np.array_equal(src[torch.arange(3)[:, None], bi].numpy(), gf["bi_first"])
    assert np.array_equal(gf["bi_second"], np.transpose(gf["bi_first"], (1, 0, 2)))


# ------------------------------------------------------------------------------------------------------------------------------
# the whole forward
# ------------------------------------------------------------------------------------------------------------------------------
def load_glue_forward(arch):
    """-> (fixture, frames uint8 [T,3,H,W], state dict with the product's prefixes, text features [K,E], oracle kwargs, out_hw)."""
    from tests._synth import GLUE_CLIP, glue_frames, glue_text
    g = np.load(os.path.join(GOLDEN, "glue_forward.npz"))
    s = [int(x) for x in g["seeds"]]
    T, H, W, K, Q, OH, OW = [int(x) for x in g["dims"]]
    Wd = synth_weights(_spec(g["spec_bb"]), s[0], "backbone.")
    Wd.update(synth_weights(_spec(g["spec_pd"]), s[1], "sem_seg_head.pixel_decoder."))
    more = [int(x) for x in g["seeds_more"]]                                  # side-adapter VIDEO decoder, AdaptedClipAdapter
    dec = {"openvis": ("spec_vdec", s[2]), "openvis_adapted": ("spec_vdec", s[2]), "openvis_online": ("spec_fdec", s[3]),
           "san": ("spec_svdec", more[0])}.get(arch, ("spec_sdec", s[4]))
    Wd.update(synth_weights(_spec(g[dec[0]]), dec[1], "sem_seg_head.predictor."))
    if arch in ("openvis", "openvis_online", "openvis_adapted"):
        if arch == "openvis_adapted":
            Wd.update(synth_weights(_spec(g["spec_acad"]), more[1], "clip_adapter."))
        else:
            Wd.update(synth_weights(_spec(g["spec_cad"]), s[5], "clip_adapter."))
        kw = dict(clip_heads=GLUE_CLIP["vision_width"] // 64, clip_resolution=GLUE_CLIP["image_resolution"])
        if arch == "openvis_adapted":
            kw.update(mask_prompt_depth=3, mask_prompt_fwd=True)
    else:
        Wd.update(synth_weights(_spec(g["spec_sad"]), s[6], "clip_adapter."))
        Wd["clip_adapter.clip_model.logit_scale"] = torch.tensor(float(g["side_logit_scale"][0]))
        kw = dict(broken_idx=3, merge_ids=(1, 2, 3), resolution=GLUE_CLIP["image_resolution"], clip_heads=4, num_queries=Q)
        if arch == "brivis":
            Wd.update(synth_weights(_spec(g["spec_res"]), s[7], "resampler."))
    # round 6 (oracle/make_golden_glue.py): the ClipAdapter towers carry peaked attention, the offline OpenVIS decoder a mask-logit gain, and
    # the two OpenVIS architectures run on a label space built from the reference's own crop embeddings (stored per architecture)
    if arch in ("openvis", "openvis_online", "openvis_adapted"):
        from openvis_amd.weights import sharpen_clip_attention
        Wd = sharpen_clip_attention(Wd, float(g["clip_qk_gain"][0]))
    if dec[0] == "spec_vdec":
        for k_ in ("weight", "bias"):
            Wd[f"sem_seg_head.predictor.mask_embed.layers.2.{k_}"] = Wd[f"sem_seg_head.predictor.mask_embed.layers.2.{k_}"] * float(g["mask_gain"][0])
    frames = glue_frames(int(g[arch + "_frame_seed"][0]))
    text = torch.from_numpy(g[arch + "_text"])
    assert arch in GLUE_SHARP or torch.equal(text, glue_text(s[8], GLUE_CLIP["embed_dim"], K))
    return g, frames, Wd, text, kw, (OH, OW)


GLUE_SHARP = ("openvis", "openvis_adapted")     # forwards whose fixture holds a SEPARATED label space: ten winners, >= 5 labels, margin >= 1e-2


ORACLE_FORWARD = {"openvis": TR.openvis_forward, "openvis_online": TR.openvis_online_forward, "san_online": TR.san_online_forward,
                  "brivis": TR.brivis_forward, "san": TR.san_forward, "openvis_adapted": TR.openvis_forward}


@pytest.mark.parametrize("arch", ["openvis", "openvis_online", "san_online", "brivis", "san", "openvis_adapted"])
def test_oracle_forward_equals_the_reference_forward(arch):
    """The reference's eval `forward` (openvis.py:47-108 / 177-242, san.py:177-283, brivis.py:105-211), run on a stub whose head /
    adapter / resampler are the reference's own modules, against the oracle's forward on the same frames and weights: the class
    probabilities handed to inference_video, the top-10 (query row, label) set, scores, entropies and the ten output masks."""
    g, frames, Wd, text, kw, out_hw = load_glue_forward(arch)
    st = {}
    with torch.no_grad():
        out = ORACLE_FORWARD[arch](frames, Wd, text, out_hw=out_hw, stages=st, **kw)
    p = arch + "_"
    assert out["image_size"] == out_hw
    assert np.abs(st["probs"].numpy() - g[p + "probs"]).max() < (1e-4 if arch in GLUE_SHARP else 1e-5)     # (an un-saturated softmax moves more per logit ulp)
    ref = {(int(r), int(l)): i for i, (r, l) in enumerate(zip(g[p + "rows"], g[p + "labels"]))}
    mine = {(int(r), int(l)): i for i, (r, l) in enumerate(zip(out["rows"], out["pred_labels"]))}
    assert set(ref) == set(mine) and len(ref) == 10
    if arch in GLUE_SHARP:                       # ... on a label space where that is a statement: >= 5 labels among the winners, the 11th candidate >= 1e-2 behind
        fl = np.sort(g[p + "probs"].reshape(-1))[::-1]
        assert len({l for _, l in ref}) >= 5 and fl[9] - fl[10] >= 1e-2 and fl[0] < 0.985 and abs(float(g[p + "margin"][0]) - (fl[9] - fl[10])) < 1e-6
    T, OH, OW = frames.shape[0], out_hw[0], out_hw[1]
    ref_masks = _bits(g[p + "masks"], (10, T, OH, OW))
    for k, i in ref.items():
        j = mine[k]
        assert abs(out["pred_scores"][j] - float(g[p + "scores"][i])) < (1e-4 if arch in GLUE_SHARP else 1e-5)
        assert abs(out["pred_entropys"][j] - float(g[p + "entropys"][i])) < 1e-4
        assert np.array_equal(out["pred_masks"][j].numpy(), ref_masks[i])
    if arch in ("openvis", "openvis_online", "openvis_adapted"):
        assert np.array_equal(st["valid"].numpy(), g[p + "valid"])
        assert np.abs(st["crop_logits"].numpy() - g[p + "crop_logits"]).max() < 1e-3          # x100 cosine logits
        assert np.array_equal(st["boxes"].numpy().astype(np.float32), g[p + "boxes"])        # the square crop boxes handed to roi_align
    if arch in ("openvis_online", "san_online"):
        assert np.array_equal(st["indices"].numpy(), g[p + "indices"])
        assert np.abs(st["pred_masks"].numpy() - g[p + "tracked_masks"].astype(np.float32)).max() < 2e-2   # stored as fp16
        assert np.abs(st["pred_embeds"].numpy() - g[p + "pred_embeds"]).max() < 1e-4         # per-frame query embeddings, the tracker's input
    if arch == "brivis":
        assert np.array_equal(st["indices"].numpy(), g[p + "indices"])
        assert np.abs(st["probs"].numpy() - g[p + "cls"]).max() < 1e-5
