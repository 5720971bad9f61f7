"""GPU: end-to-end and stage-wise parity of the OpenVIS R50 path (HIP kernels through the C ABI) against the CPU
oracle (oracle/torch_ref.py) on a small synthetic clip with seeded random weights of the real architecture
(ResNet-50 + 6-layer MSDeformAttn pixel decoder + 9-layer masked-attention decoder; a small CLIP ViT keeps the
oracle fast).  Tolerances are stated per stage; thresholded quantities are compared by IoU / exact match rate."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CLIP_ARCH = dict(width=256, layers=2, heads=4, patch=16, resolution=64, embed_dim=64)
T, H, W, K = 2, 90, 120, 7


def _frames(seed=0):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(T, 3, H, W, generator=g) * 255
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for t in range(T):
        blob = 120 * torch.exp(-(((yy - 40 - 5 * t) / 18.0) ** 2 + ((xx - 60 + 7 * t) / 25.0) ** 2))
        base[t] = (base[t] * 0.4 + blob).clamp(0, 255)
    return base.to(torch.uint8)


# (dense-path policy, CLIP tower operand dtype): all-f32 parity mode, f32 dense + fp16 CLIP, and the reference's own
# autocast policy ("mixed": backbone + decoder GEMM operands fp16 / f32 accumulate, pixel decoder f32, CLIP fp16)
@pytest.fixture(scope="module", params=[("fp32", "fp32", "fp32"), ("fp32", "fp16", "fp32"), ("mixed", "fp16", "fp32"),
                                        ("mixed", "fp16", "fp16")],
                ids=["f32", "f32+clip16", "mixed", "mixed+decoder16"])
def case(request):
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    from oracle import torch_ref as TR

    spec = (weights.resnet50_spec() + weights.pixel_decoder_spec() + weights.video_decoder_spec() +
            weights.clip_visual_spec(**CLIP_ARCH))
    sd = weights.random_init(spec, seed=7)
    policy, clip_prec, dec_prec = request.param
    cfg = config.get_cfg()
    cfg.MODEL.PRECISION = policy
    cfg.MODEL.MASK_FORMER.DECODER_PRECISION = dec_prec         # "fp16": the autocast behaviour of the decoder
    model = config.build_model(cfg)
    model.clip_adapter = ClipAdapter("tiny", arch=CLIP_ARCH, precision=clip_prec)
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_val").set(thing_classes=names)
    g = torch.Generator().manual_seed(1)
    # class embeddings close to a common direction so the x100 cosine logits differ by O(1) (an un-saturated softmax;
    # independent random unit vectors saturate every score at exactly 1.0 and make top-k a pure tie-break)
    base = torch.randn(1, CLIP_ARCH["embed_dim"], generator=g)
    text = torch.nn.functional.normalize(base + 0.05 * torch.randn(K, CLIP_ARCH["embed_dim"], generator=g), dim=-1)
    model.clip_adapter.set_text_features(names, text)
    frames = _frames()
    st_gpu = {}
    out_gpu = model([{"image": [f for f in frames], "dataset_name": "synthetic_val"}], stages=st_gpu)
    torch.cuda.synchronize()
    st_ref = {}
    with torch.no_grad():
        out_ref = TR.openvis_forward(frames, sd, text, stages=st_ref, clip_heads=CLIP_ARCH["heads"],
                                     clip_resolution=CLIP_ARCH["resolution"])
    return dict(out_gpu=out_gpu, out_ref=out_ref, st_gpu=st_gpu, st_ref=st_ref, precision=clip_prec, policy=policy,
                tag=f"{policy}+clip_{clip_prec}" + ("+decoder_fp16" if (policy == "mixed" and dec_prec == "fp16") else ""))


def _report(case, key, val):
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/parity_report.json"
    d = json.load(open(path)) if os.path.exists(path) else {}
    d.setdefault(case["tag"], {})[key] = val
    json.dump(d, open(path, "w"), indent=1)


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def test_a1_preprocess(case):
    g = case["st_gpu"]["images"][..., :3].permute(0, 3, 1, 2).cpu()
    assert torch.equal(g, case["st_ref"]["images"])                       # integer inputs, same fp32 ops: bit-exact
    assert case["st_gpu"]["images"][..., 3].abs().max().item() == 0


def test_a2_backbone(case):
    tol = 1e-4 if case["policy"] == "fp32" else 5e-3                     # f32 exact-MFMA vs fp16-operand autocast policy
    for k in ("res2", "res3", "res4", "res5"):
        g = case["st_gpu"]["features"][k].permute(0, 3, 1, 2).cpu()
        assert _rel(g, case["st_ref"]["feats"][k]) < tol, k              # different summation order / folded BN


def test_a3_a8_masks(case):
    g = case["st_gpu"]["pred_masks"].cpu()
    r = case["st_ref"]["pred_masks"]
    assert g.shape == r.shape
    assert _rel(g, r) < (5e-3 if case["policy"] == "fp32" else 3e-2)
    agree = ((g > 0) == (r > 0)).float().mean().item()
    inter = ((g > 0) & (r > 0)).sum().item()
    union = ((g > 0) | (r > 0)).sum().item()
    _report(case, "pred_masks", dict(rel_err=_rel(g, r), sign_agree=agree, iou=inter / max(union, 1)))
    assert agree > 0.999 and inter / max(union, 1) > 0.999, (agree, inter / max(union, 1))


def test_a10_valid_flags_and_boxes(case):
    vg, vr = case["st_gpu"]["valid"], case["st_ref"]["valid"].numpy()
    assert (vg == vr).mean() > 0.99
    both = vg & vr
    # boxes of crops valid on both sides (oracle keeps x1+1/y1+1 square boxes; compare the top-left anchor + side)
    crops = case["st_gpu"]["crops"]
    gb = {(int(c[0]), int(c[1])): c[2:] for c in crops}
    rb = case["st_ref"]["boxes"].numpy()
    ids = np.argwhere(vr)
    exact = 0
    for (t, q), b in zip(ids, rb):
        if both[t, q]:
            x0, y0, x1, y1 = gb[(t, q)]
            side = max(x1 + 1 - x0, y1 + 1 - y0)
            exact += int(b[0] == x0 and b[1] == y0 and b[2] == x0 + side and b[3] == y0 + side)
    assert exact / max(int(both.sum()), 1) > 0.97


def test_a12_clip_logits_and_probs(case):
    vg, vr = case["st_gpu"]["valid"], case["st_ref"]["valid"].numpy()
    lg = case["st_gpu"]["crop_logits"].cpu().numpy()
    lr = case["st_ref"]["crop_logits"].numpy()
    ig = {tuple(x): i for i, x in enumerate(np.argwhere(vg))}
    ir = {tuple(x): i for i, x in enumerate(np.argwhere(vr))}
    common = [k for k in ig if k in ir]
    d = np.array([np.abs(lg[ig[k]] - lr[ir[k]]).max() for k in common])
    # cosine logits x100: north-star tolerance 1e-3 on the cosine -> 1e-1 on the x100 logits.
    # fp32 tower: ~1e-3 on the logits; fp16 GEMM operands (the reference's GPU dtype): within the 1e-1 bound.
    _report(case, "clip_logit_abs_err", dict(median=float(np.median(d)), p99=float(np.quantile(d, 0.99)), max=float(d.max())))
    # the BOUND: every crop whose box is identical on both sides is within 1e-3 on the cosine (1e-1 on the x100 logits);
    # crops whose box moved with a boundary pixel of their mask see a different picture and are only counted
    from tests._logits import logit_errors_by_box
    d_same, d_diff = logit_errors_by_box(case["st_gpu"], case["st_ref"])
    _report(case, "clip_logit_abs_err_same_box", dict(n=int(len(d_same)), max=float(d_same.max()), n_other_box=int(len(d_diff))))
    assert len(d_same) >= 0.95 * (len(d_same) + len(d_diff)) and len(d_same) > 0
    assert d_same.max() <= (1e-2 if case["policy"] == "fp32" and case["precision"] == "fp32" else 1e-1), d_same.max()


def test_a16_video_output(case):
    og, orf = case["out_gpu"], case["out_ref"]
    assert og["image_size"] == orf["image_size"] == (H, W)
    sg = {(q, l): (s, e) for q, l, s, e in zip(og["pred_queries"], og["pred_labels"], og["pred_scores"], og["pred_entropys"])}
    rows_ref = case["st_ref"]["valid"].any(0).nonzero()[:, 0].tolist()
    sr = {(rows_ref[r], l): (s, e) for r, l, s, e in zip(orf["rows"], orf["pred_labels"], orf["pred_scores"], orf["pred_entropys"])}
    common = set(sg) & set(sr)
    assert len(common) >= 8, (sorted(sg), sorted(sr))                   # top-10 as a set keyed by (query, label)
    tol_s, tol_e = (2e-3, 2e-2) if (case["precision"] == "fp32" and case["policy"] == "fp32") else (2e-2, 1e-1)
    _report(case, "top10", dict(common=len(common), max_score_err=max(abs(sg[k][0] - sr[k][0]) for k in common),
                                max_entropy_err=max(abs(sg[k][1] - sr[k][1]) for k in common)))
    for k in common:
        assert abs(sg[k][0] - sr[k][0]) < tol_s and abs(sg[k][1] - sr[k][1]) < tol_e
    # masks of the common detections
    mg = {(q, l): m for q, l, m in zip(og["pred_queries"], og["pred_labels"], og["pred_masks"])}
    mr = {(rows_ref[r], l): m for r, l, m in zip(orf["rows"], orf["pred_labels"], orf["pred_masks"])}
    for k in common:
        a, b = mg[k], mr[k]
        assert a.shape == b.shape == (T, H, W)
        union = (a | b).sum().item()
        assert union == 0 or (a & b).sum().item() / union > 0.999
