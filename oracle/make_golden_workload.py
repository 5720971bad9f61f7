"""TEST INFRASTRUCTURE ONLY -- golden vectors at WORKLOAD size for BASELINE.json configs[2], [3], [4]:
the f32 CPU oracle (oracle/torch_ref.py, pinned against the reference's own modules by tests/test_oracle_*.py) is run
ONCE in the build container on seeded synthetic inputs, and what the GPU tests compare against is committed as data:

  tests/golden/c2_openvis_720p_5f.npz    OpenVIS R50 + ClipAdapter ViT-B/16, the bench workload at full size: 5 frames of 720x1280, 482 classes
  tests/golden/c2_sharp_classes.npz      the same clip, synthetic CLIP tower with peaked attention + a label space built from the oracle's own
                                         crop embeddings: a classification result that can differ (`c2s`; oracle/fixtures.py)
  tests/golden/c2_autocast_backbone.npz  the same clip's mask sign bits with the backbone in the reference's GPU arithmetic (autocast: every op's
                                         output rounded to fp16; `c2a`), and their distance to the f32 oracle's
  tests/golden/c3_san_online_720p.npz    SANOnline R50 + SideAdapter ViT-B/16, 5 frames of 720x1280 (the config's T)
  tests/golden/c4_brivis_720p_36f.npz    BriVIS R50, ONE 36-frame 720p clip (linker over all 36 frames, resampler, heads)
  tests/golden/c5_brivis_swinl_1080p.npz BriVIS Swin-L (embed 192, depths 2/2/18/2, window 12) + SideAdapter ViT-L/14@336,
                                         3 frames of 1080x1920 (tracker + temporal resampler active)
  tests/golden/c5_brivis_swinl_1080p_36f.npz  the same model on the config's full 36 frames (`c5f`: ~20 min, ~30 GB)

Inputs are NOT stored: frames = bench.synth_frames(T, H, W, seed), weights = weights.random_init(spec, seed 42), text =
bench.synth_text(40, E) -- all seeded torch CPU generators, identical on the GPU box (same image).  Stored per case: tracker
indices [T,Q], class probabilities [Q,K], per-frame logits of a few queries, the top-10 (row, label, score), positive-pixel
counts of every (frame, query) mask, the sign bits of the mask logits (np.packbits) for all queries on a frame subset, and next to
them the bitmaps of the pixels whose oracle logit is within eps of zero (`ambiguous`): the only places a GPU mask bit may differ.

  python oracle/make_golden_workload.py c3 c4 c5        (c4: ~10 min and ~12 GB on 8 cores; frames go through the oracle in
                                                         chunks of 4 -- every per-frame stage is frame-independent)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

import bench
from openvis_amd import weights
from openvis_amd.modeling.clip_adapter.adapter import _CLIP_ARCH
from oracle import torch_ref as TR

GOLDEN = os.path.join(ROOT, "tests", "golden")
K, Q = 40, 100


def image_outputs_chunked(frames, W, text, chunk, **kw):
    """san_online_image_outputs over frame chunks (identical results: nothing before the linker mixes frames)."""
    outs = []
    for b in range(0, frames.shape[0], chunk):
        t0 = time.time()
        with torch.no_grad():
            o = TR.san_online_image_outputs(frames[b:b + chunk], W, text, **kw)
        outs.append(o)
        print(f"  frames {b}..{min(b + chunk, frames.shape[0]) - 1}: {time.time() - t0:.1f} s", flush=True)
    io = dict(outs[0])
    for k in ("pred_masks",):
        io[k] = torch.cat([o[k] for o in outs], dim=2)                      # [1,Q,T,h,w]
    for k in ("pred_embeds", "pred_logits"):
        io[k] = torch.cat([o[k] for o in outs], dim=1)                      # [1,T,Q,*]
    for k in ("mask_feats", "attn_feats", "images"):
        io[k] = torch.cat([o[k] for o in outs], dim=0)
    io.pop("class_attn_biases", None)
    cb = [o["clip_bk"] for o in outs]
    io["clip_bk"] = tuple(torch.cat([c[i] for c in cb], dim=(1 if i == 0 else 0)) for i in range(len(cb[0])))
    return io


def pack(mask_logits):
    """sign bits of [..., h, w] logits -> uint8, 8 pixels per byte."""
    return np.packbits((mask_logits > 0).numpy().astype(np.uint8), axis=-1)


AMBIG_EPS = (1e-4, 1e-3, 3e-2)


def ambiguous(mask_logits):
    """"Bit-exact masks" as a checkable statement: next to the sign bits, the packed bitmaps of the pixels whose ORACLE logit lies within
    eps of zero, for eps in AMBIG_EPS.  A GPU mask bit may differ from the oracle's only inside such a set: 1e-3 for the f32-class
    SAN-family paths C3 / C4 / C5 (measured: 21-60 differing bits per case, 0-4 of them beyond 1e-4, none beyond 1e-3), 3e-2 for C2
    (fp16-operand backbone = the reference's autocast: logits move by up to 2e-2; with an f32 backbone 116-126 bits differ, 5 beyond 1e-3).
    The set sizes say how sharp the statement is: ~400-900 / 4-9 k / 130-260 k of the 17-29 M pixels of a case for the three eps."""
    a = mask_logits.abs()
    d = {"ambig_eps": np.asarray(AMBIG_EPS, np.float64)}
    for i, eps in enumerate(AMBIG_EPS):
        d[f"ambig_bits_{i}"] = np.packbits((a < eps).numpy().astype(np.uint8), axis=-1)
    return d


def save(name, **arrays):
    path = os.path.join(GOLDEN, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


def topk_arrays(res):
    return dict(top_rows=np.asarray(res["rows"], np.int32), top_labels=np.asarray(res["pred_labels"], np.int32),
                top_scores=np.asarray(res["pred_scores"], np.float32), top_entropys=np.asarray(res["pred_entropys"], np.float32))


def c2():
    """BASELINE.json configs[1] at its FULL size: OpenVIS R50, one 5-frame 720p clip (bench.py's clip 0), 100 queries, 482 classes."""
    T, K2 = 5, 482
    sd = weights.random_init(weights.openvis_spec("r50", None, Q), seed=42)
    frames = bench.synth_frames(T, 720, 1280, 1000, "cpu")
    text = bench.synth_text(K2, 512, spread=0.25)              # separated classes, as tests/test_c2_720p_gpu.py
    st = {}
    with torch.no_grad():
        res = TR.openvis_forward(frames, sd, text, stages=st)
    pm = st["pred_masks"][0]                                               # [Q,T,h,w]
    valid = st["valid"].numpy()
    margin = np.abs(pm.numpy()).min()
    print(f"  valid crops {int(valid.sum())}, smallest |mask logit| {margin:.2e}", flush=True)
    save("c2_openvis_720p_5f.npz", mask_bits=pack(pm), mask_shape=np.array(pm.shape), valid=valid.astype(np.uint8),
         boxes=st["boxes"].numpy().astype(np.int32), crop_logits=st["crop_logits"].numpy().astype(np.float32), probs=st["probs"].numpy(),
         mask_counts=(pm > 0).sum(dim=(-1, -2)).numpy().astype(np.int32), **ambiguous(pm), **topk_arrays(res))


def c2s():
    """C2 again with a label space on which the classification can FAIL (oracle/fixtures.py): the synthetic tower gets peaked attention
    (weights.sharpen_clip_attention: crop embeddings differ between queries; masks are those of c2 -- the CLIP weights do not reach them),
    the text rows are built from the oracle's own per-query embeddings.  Stored: the parts of the text, the oracle's crop embeddings
    [500, 512] (expected logits = 100 E text^T), class probabilities, the top-10 and the pixel counts of its ten output masks."""
    from oracle import fixtures as FX
    T, K2 = 5, 482
    sd = weights.sharpen_clip_attention(weights.random_init(weights.openvis_spec("r50", None, Q), seed=42))
    frames = bench.synth_frames(T, 720, 1280, 1000, "cpu")
    st = {}
    with torch.no_grad():
        TR.openvis_forward(frames, sd, bench.synth_text(K2, 512, spread=0.25), stages=st)
        E, valid = st["crop_embeds"], st["valid"]
        rows, Mq = FX.per_query_mean(E, valid)
        parts, rep = FX.sharp_parts(Mq, K2)
        text = FX.text_from_parts(parts, K2)
        crop_logits = 100.0 * E @ text.T
        mask_pred = F.interpolate(st["pred_masks"][0], size=st["images"].shape[-2:], mode="bilinear", align_corners=False)
        probs, vmasks, _ = TR.aggregate_crop_logits(crop_logits, valid, mask_pred)
        res = TR.inference_video(Q, K2, probs, vmasks, (720, 1280), 720, 1280)
    print("  label space:", rep, flush=True)
    assert rep["margin"] >= 1e-2 and rep["distinct_labels"] >= 5
    assert sorted((r, l) for r, l in zip(res["rows"], res["pred_labels"])) == rep["top"]
    save("c2_sharp_classes.npz", crop_embeds=E.numpy().astype(np.float32), valid=valid.numpy().astype(np.uint8), boxes=st["boxes"].numpy().astype(np.int32),
         probs=probs.numpy(), rows=np.asarray(rows, np.int32), margin=np.asarray([rep["margin"]]),
         top_mask_counts=np.asarray([int(m.sum()) for m in res["pred_masks"]], np.int64), **FX.parts_arrays(parts), **topk_arrays(res))


def c2a():
    """C2's masks with the backbone in the reference's own GPU arithmetic (TR.resnet50_autocast: every conv / BN / add output rounded to
    fp16, as `autocast` does at train_net.py:241), everything behind it f32 as in c2.  Stored: the sign bits of the mask logits, and how
    many of them differ from the f32 oracle's (c2_openvis_720p_5f.npz) in total and outside its |logit| < eps sets -- the envelope the
    reference's own arithmetic has against f32, which the product's fp16-operand backbone is measured against (tests/test_c2_720p_gpu.py)."""
    T = 5
    sd = weights.random_init(weights.openvis_spec("r50", None, Q), seed=42)
    frames = bench.synth_frames(T, 720, 1280, 1000, "cpu")
    with torch.no_grad():
        images, _ = TR.preprocess([f for f in frames])
        feats = TR.resnet50_autocast(images, sd)
        mask_features, _, ms = TR.pixel_decoder(feats, sd)
        _, pred_masks = TR.video_decoder(ms, mask_features, sd)
    pm = pred_masks[0]
    g = np.load(os.path.join(GOLDEN, "c2_openvis_720p_5f.npz"))
    ref = np.unpackbits(g["mask_bits"], axis=-1)[..., : int(g["mask_shape"][-1])].astype(bool)
    got = (pm > 0).numpy()
    diff = got != ref
    outside = []
    for i, eps in enumerate(g["ambig_eps"]):
        amb = np.unpackbits(g[f"ambig_bits_{i}"], axis=-1)[..., : ref.shape[-1]].astype(bool)
        outside.append(int((diff & ~amb).sum()))
    inter, union = (got & ref).sum(axis=(1, 2, 3)).astype(np.float64), (got | ref).sum(axis=(1, 2, 3)).astype(np.float64)
    iq = np.where(union > 0, inter / np.maximum(union, 1), 1.0)
    print(f"  autocast-arithmetic backbone vs f32 oracle: {int(diff.sum())} of {diff.size} mask bits differ, outside the |logit| < "
          f"{g['ambig_eps'].tolist()} sets: {outside}; per-query IoU min {iq.min():.5f}; {int((~diff).all(axis=(1, 2, 3)).sum())} of {Q} masks bit-identical", flush=True)
    save("c2_autocast_backbone.npz", mask_bits=pack(pm), mask_shape=np.array(pm.shape), n_diff_vs_f32=np.asarray([int(diff.sum())], np.int64),
         outside_vs_f32=np.asarray(outside, np.int64), ambig_eps=g["ambig_eps"], iou_min_vs_f32=np.asarray([iq.min()]))


def c3():
    T = 5
    sd = weights.random_init(weights.san_spec("r50", None, Q), seed=42)
    frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
    text = bench.synth_text(K, 512)
    st = {}
    with torch.no_grad():
        res = TR.san_online_forward(frames, sd, text, stages=st)
    pm = st["pred_masks"][0]                                               # [Q,T,h,w]
    save("c3_san_online_720p.npz", indices=st["indices"].numpy().astype(np.int16), probs=st["probs"].numpy(),
         logits=st["pred_logits"][0].numpy().astype(np.float32), mask_bits=pack(pm), mask_shape=np.array(pm.shape),
         mask_counts=(pm > 0).sum(dim=(-1, -2)).numpy().astype(np.int32), **ambiguous(pm), **topk_arrays(res))


def brivis_case(name, frames, sd, text, chunk, keep_frames, **kw):
    nq = kw.get("num_queries", Q)
    io = image_outputs_chunked(frames, sd, text, chunk, **kw)
    images, (H, Wd) = io["images"], io["image_size"]
    with torch.no_grad():
        idx, frame_embeds = TR.video_match_via_embeds(io["pred_embeds"][0])
        x = TR.resampler_forward(frame_embeds.unsqueeze(0), sd)
        masks, biases, emb = TR.resampler_heads(x, io["mask_feats"], io["attn_feats"], sd)
        sos = TR.san_post_encode_image(io["clip_bk"], biases, sd, broken_idx=kw.get("broken_idx", 9), num_sos=nq)
        logits = TR.san_cal_sim_logits(io["text_feats"], sos, sd)                        # [T,Q,K+1]
        probs = F.softmax(logits.mean(dim=0), dim=-1)[:, :-1]
        pred_masks = masks.permute(1, 0, 2, 3)                                           # [Q,T,h,w]
        # top-10 on the low-res masks is what the GPU test compares; the full-size output masks of the SUBSET frames only
        mask_pred = F.interpolate(pred_masks[:, keep_frames], size=images.shape[-2:], mode="bilinear", align_corners=False)
        res = TR.inference_video(nq, text.shape[0], probs, mask_pred, (H, Wd), H, Wd)
    save(name, indices=idx.numpy().astype(np.int16), probs=probs.numpy(), logits_subset=logits[keep_frames].numpy().astype(np.float32),
         keep_frames=np.asarray(keep_frames, np.int32), mask_bits=pack(pred_masks[:, keep_frames]),
         mask_shape=np.array(pred_masks[:, keep_frames].shape), mask_counts=(pred_masks > 0).sum(dim=(-1, -2)).numpy().astype(np.int32),
         pred_embeds_checksum=emb.double().abs().sum(dim=(1, 2)).numpy(), **ambiguous(pred_masks[:, keep_frames]), **topk_arrays(res))


def c4():
    T = 36
    sd = weights.random_init(weights.brivis_spec("r50", None, Q), seed=42)
    frames = bench.synth_frames(T, 720, 1280, 1000, "cpu")               # bench.py --model brivis, clip 0
    text = bench.synth_text(K, 512)
    brivis_case("c4_brivis_720p_36f.npz", frames, sd, text, 4, [0, 17, 35])


def c5():
    arch = _CLIP_ARCH["ViT-L/14@336px"]
    a = weights.SWIN_ARCH["swin_l"]
    sd = weights.random_init(weights.brivis_spec("swin_l", arch, Q), seed=42)
    frames = bench.synth_frames(3, 1080, 1920, 1000, "cpu")              # 3 frames: the linker and the resampler's temporal attention do real work
    text = bench.synth_text(K, arch["embed_dim"])
    bb = lambda images, W: TR.swin(images, W, a["embed_dim"], a["depths"], a["num_heads"], a["window"])
    brivis_case("c5_brivis_swinl_1080p.npz", frames, sd, text, 1, [0, 2], broken_idx=21, merge_ids=(6, 12, 18), resolution=336,
                clip_heads=arch["width"] // 64, num_queries=Q, backbone_fn=bb)


def c5f():
    """BASELINE.json configs[4] at its FULL T: 36 frames of 1080x1920 (one frame at a time through the oracle: ~25 s per frame on 8 cores,
    ~30 GB peak for the 36-frame mask / bias einsums of the resampler heads); masks stored for frames 0 / 17 / 35, pixel counts for all."""
    arch = _CLIP_ARCH["ViT-L/14@336px"]
    a = weights.SWIN_ARCH["swin_l"]
    sd = weights.random_init(weights.brivis_spec("swin_l", arch, Q), seed=42)
    frames = bench.synth_frames(36, 1080, 1920, 1000, "cpu")
    text = bench.synth_text(K, arch["embed_dim"])
    bb = lambda images, W: TR.swin(images, W, a["embed_dim"], a["depths"], a["num_heads"], a["window"])
    brivis_case("c5_brivis_swinl_1080p_36f.npz", frames, sd, text, 1, [0, 17, 35], broken_idx=21, merge_ids=(6, 12, 18), resolution=336,
                clip_heads=arch["width"] // 64, num_queries=Q, backbone_fn=bb)


if __name__ == "__main__":
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for case in sys.argv[1:] or ["c3", "c4", "c5"]:
        t0 = time.time()
        print(f"== {case}", flush=True)
        {"c2": c2, "c2s": c2s, "c2a": c2a, "c3": c3, "c4": c4, "c5": c5, "c5f": c5f}[case]()
        print(f"== {case} done in {time.time() - t0:.0f} s", flush=True)
