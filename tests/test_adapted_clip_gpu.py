"""GPU: the mask-adapted CLIP path (AdaptedClipAdapter, mask_adapted_adapter.py:35-148 -> VisionTransformer.forward(x, m),
mask_adapted_clip/model.py:327-362) and the Bg* adapters' non-object row, against the reference's golden vector and
the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests._synth import synth_inputs, synth_weights

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _spec(arr):
    return [(k, tuple(s)) for k, s in json.loads(bytes(arr.tolist()).decode())]


def _im2col(x, ps, row_len):
    """[M,3,R,R] -> the patch matrix layout of ovis_clip_crop_patches: row (m, py, px), column c*ps*ps + iy*ps + ix."""
    M, C, R, _ = x.shape
    G = R // ps
    a = x.view(M, C, G, ps, G, ps).permute(0, 2, 4, 1, 3, 5).reshape(M * G * G, C * ps * ps)
    return F.pad(a, (0, row_len - a.shape[1])).contiguous()


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("fp16", 2e-2)])
def test_mask_prompt_tower_matches_reference_golden(precision, tol):
    from openvis_amd import ops
    from openvis_amd.modeling.clip_adapter.adapter import ClipVisual
    g = np.load(os.path.join(GOLDEN, "clip_visual_mask_prompt.npz"))
    prefix = "clip_adapter.clip_model.visual."
    Wd = synth_weights(_spec(g["spec"]), int(g["seeds"][0]), prefix)
    x = synth_inputs([(4, 3, 64, 64)], int(g["seeds"][1]))[0]
    m = torch.from_numpy(g["mask"])
    vis = ClipVisual(width=256, layers=4, heads=4, patch=16, resolution=64, embed_dim=64, precision=precision,
                     mask_prompt_depth=3).load_state_dict(Wd, prefix, "cuda")
    A = _im2col(x, 16, ops.patch_row_len(16)).cuda()
    if precision == "fp16":
        A = A.half()
    patch_open = torch.ceil(F.avg_pool2d(m, 16, 16)).reshape(4, 16).to(torch.uint8).cuda()
    assert 0 < int(patch_open.sum()) < patch_open.numel()
    out = vis.forward_patches(A, 4, patch_open).cpu()
    ref = torch.from_numpy(g["out"])
    assert (out - ref).abs().max().item() / ref.abs().max().item() < tol
    plain = vis.forward_patches(A, 4).cpu()                              # the prompt is not a no-op on this fixture
    assert (plain - ref).abs().max().item() / ref.abs().max().item() > 10 * tol


def test_mask_prompt_select_kernel_exact():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(3)
    M, L, C = 5, 9, 64
    x = torch.randn(M, L + 1, C, generator=g)
    emb = torch.randn(L, C, generator=g)
    op = (torch.rand(M, L, generator=g) > 0.4).to(torch.uint8)
    ref = x.clone()
    ref[:, 1:] = torch.where(op.bool()[..., None], x[:, 1:], emb[None])
    xd = x.cuda()
    ops.mask_prompt_select(xd, op.cuda(), emb.cuda(), 1)
    assert torch.equal(xd.cpu(), ref)
    # single shared row (mask_embedding.shape[1] == 1, model.py:335-336) and first_token 0 on the [M*L, C] patch matrix
    y = torch.randn(M * L, C, generator=g)
    ref2 = torch.where(op.bool().reshape(-1, 1), y, emb[:1])
    yd = y.cuda()
    ops.mask_prompt_select(yd, op.cuda(), emb[:1].contiguous().cuda(), 0)
    assert torch.equal(yd.cpu(), ref2)


def _adapter_case(seed=5):
    g = torch.Generator().manual_seed(seed)
    T, Q, H, W = 2, 6, 70, 90
    Hp, Wp = 96, 96
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8)
    masks = torch.randn(Q, T, Hp // 4, Wp // 4, generator=g) * 3
    # wide, flat objects near the right / bottom border: their SQUARE boxes reach beyond the padded frame, so whole
    # patches of the mask region are exactly 0 (the prompt's closed patches)
    masks[0] = -8.0
    masks[0, :, 2:5, 2:23] = 8.0
    masks[1] = -8.0
    masks[1, :, 15:23, 19:22] = 8.0
    return frames, masks, (Hp, Wp)


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-4), ("fp16", 3e-2)])
@pytest.mark.parametrize("fwd", [True, False])
def test_adapted_clip_adapter_vs_oracle(precision, tol, fwd):
    from openvis_amd import ops, weights
    from openvis_amd.modeling.clip_adapter import AdaptedClipAdapter
    from openvis_amd.modeling.clip_adapter.adapter import PIXEL_MEAN, PIXEL_STD
    from oracle import torch_ref as TR
    arch = dict(width=256, layers=4, heads=4, patch=16, resolution=64, embed_dim=64)
    sd = weights.random_init(weights.clip_visual_spec(**arch, mask_prompt_depth=3), seed=33)
    ad = AdaptedClipAdapter("tiny", 3, fwd, arch=arch, precision=precision).load_state_dict(sd, "clip_adapter.", "cuda")
    frames, masks, (Hp, Wp) = _adapter_case()
    names = [f"c{i}" for i in range(6)]
    text = F.normalize(torch.randn(6, 64, generator=torch.Generator().manual_seed(1)), dim=-1)
    ad.set_text_features(names, text)
    logits, valid, crops = ad(frames.cuda(), names, masks.cuda(), (Hp, Wp))
    up = F.interpolate(masks, size=(Hp, Wp), mode="bilinear", align_corners=False)
    part = up.sigmoid().transpose(0, 1).contiguous()
    with torch.no_grad():
        regions, v2, _, mregions = TR.clip_crops(frames, part, resolution=64, return_mask_regions=True)
        feat = TR.clip_encode_image(regions, sd, resolution=64, heads=4, mask_regions=mregions if fwd else None,
                                    mask_prompt_depth=3)
        ref = 100.0 * feat @ text.T
    assert (valid == v2.numpy()).all()
    assert (logits.cpu() - ref).abs().max().item() < tol * 100, (logits.cpu() - ref).abs().max().item()
    # the pooled mask itself: bit-exact 0/1 pattern, with closed patches present
    _, patch_open = ops.clip_crop_patches_masked(frames.cuda(), masks.cuda(), torch.from_numpy(crops).cuda(), Hp, Wp, 64, 16,
                                                 PIXEL_MEAN, PIXEL_STD)
    ref_open = torch.ceil(F.avg_pool2d(mregions.half().float(), 16, 16)).reshape(len(crops), -1)
    assert torch.equal(patch_open.cpu().float(), ref_open)
    assert 0 < int(ref_open.sum()) < ref_open.numel()


def test_mask_prompt_closes_background_patches_like_the_fp16_reference():
    """Real checkpoints put background logits near -20: the reference's fp16 mask regions (mask_adapted_adapter.py:113)
    are exactly 0 there, so ceil(avg_pool(m)) closes those patches (model.py:332-338).  An f32 sigmoid never is 0: a kernel
    that tests `mask > 0` in f32 opens every in-frame patch and the mask prompt degenerates to the plain ViT.  Expected
    pattern: roi_align of the fp16-rounded soft mask, rounded to fp16 again (the reference's tensors), pooled, ceil'ed."""
    from openvis_amd import ops
    from openvis_amd.modeling.clip_adapter.adapter import PIXEL_MEAN, PIXEL_STD
    from oracle import torch_ref as TR
    g = torch.Generator().manual_seed(11)
    T, Q, H, W, Hp, Wp = 2, 4, 96, 128, 96, 128
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8)
    masks = torch.full((Q, T, Hp // 4, Wp // 4), -22.0) + torch.randn(Q, T, Hp // 4, Wp // 4, generator=g)
    masks[0, :, 3:9, 4:20] = 8.0                       # four objects of different shapes inside the frame
    masks[1, :, 10:22, 24:29] = 7.0
    masks[2, :, 2:20, 2:6] = 9.0
    masks[3, :, 12:15, 8:27] = 6.0
    up = F.interpolate(masks, size=(Hp, Wp), mode="bilinear", align_corners=False)
    part = up.sigmoid().transpose(0, 1).contiguous()
    _, valid, boxes, m32 = TR.clip_crops(frames, part, resolution=64, return_mask_regions=True)
    _, _, _, m16 = TR.clip_crops(frames, part.half().float(), resolution=64, return_mask_regions=True)
    ref_open = torch.ceil(F.avg_pool2d(m16.half().float(), 16, 16)).reshape(m16.shape[0], -1)
    f32_open = torch.ceil(F.avg_pool2d(m32, 16, 16)).reshape(m32.shape[0], -1)
    bx = ops.mask_bbox(masks.cuda(), Hp, Wp).cpu().numpy()                # as ClipAdapter.preprocess_boxes
    ok = bx[..., 2] >= 0
    assert (ok == valid.numpy()).all()
    crops = torch.from_numpy(np.concatenate([np.argwhere(ok), bx[ok]], axis=1).astype(np.int32))
    _, patch_open = ops.clip_crop_patches_masked(frames.cuda(), masks.cuda(), crops.cuda(), Hp, Wp, 64, 16, PIXEL_MEAN, PIXEL_STD)
    got = patch_open.cpu().float()
    assert torch.equal(got, ref_open), (got != ref_open).sum().item()
    assert int((f32_open != ref_open).sum()) >= 8      # the f32 rule opens background patches the reference closes
    assert 0 < int(ref_open.sum()) < ref_open.numel()
    # and the oracle's tower follows the same rule
    assert torch.equal(torch.ceil(F.avg_pool2d(m32.half().float(), 16, 16)).reshape(m32.shape[0], -1), ref_open)


def test_bg_adapters_append_the_non_object_row():
    from openvis_amd import weights
    from openvis_amd.modeling.clip_adapter import ADAPTER_REGISTER, BgAdaptedClipAdapter, BgClipAdapter
    assert sorted(ADAPTER_REGISTER) == ["AdaptedClipAdapter", "BgAdaptedClipAdapter", "BgClipAdapter", "ClipAdapter"]
    arch = dict(width=256, layers=2, heads=4, patch=16, resolution=64, embed_dim=64)
    frames, masks, (Hp, Wp) = _adapter_case()
    names = [f"c{i}" for i in range(6)]
    text = F.normalize(torch.randn(6, 64, generator=torch.Generator().manual_seed(1)), dim=-1)
    for cls, kw, depth in ((BgClipAdapter, {}, 0), (BgAdaptedClipAdapter, dict(mask_prompt_depth=2, mask_prompt_fwd=True), 2)):
        sd = weights.random_init(weights.clip_visual_spec(**arch, mask_prompt_depth=depth) +
                                 [("clip_adapter.non_object_embedding", (1, 64))], seed=35)
        ad = cls("tiny", arch=arch, precision="fp32", **kw).load_state_dict(sd, "clip_adapter.", "cuda")
        ad.set_text_features(names, text)
        logits, valid, crops = ad(frames.cuda(), names, masks.cuda(), (Hp, Wp))
        assert logits.shape == (len(crops), 7)
        tf = ad.encode_text(names).cpu()                                  # adapter.py:157-161
        bg = sd["clip_adapter.non_object_embedding"]
        assert torch.allclose(tf[:6], text, atol=1e-6) and torch.allclose(tf[6:], bg / bg.norm(dim=-1, keepdim=True), atol=1e-6)


def test_openvis_with_adapted_clip_adapter_end_to_end():
    """OpenVIS built from a config that names AdaptedClipAdapter, against the oracle's forward with the mask prompt."""
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter import AdaptedClipAdapter
    from oracle import torch_ref as TR
    arch = dict(width=256, layers=3, heads=4, patch=16, resolution=64, embed_dim=64)
    cfg = config.get_cfg()
    cfg.MODEL.PRECISION = "fp32"
    cfg.MODEL.CLIP_ADAPTER.NAME = "AdaptedClipAdapter"
    cfg.MODEL.CLIP_ADAPTER.MASK_PROMPT_DEPTH = 2
    model = config.build_model(cfg)
    assert isinstance(model.clip_adapter, AdaptedClipAdapter) and model.clip_adapter.mask_prompt_depth == 2
    assert any(k.endswith("visual.mask_embedding") for k, _ in weights.spec_for_cfg(cfg))
    sd = weights.random_init(weights.openvis_spec("r50", arch, 100, "AdaptedClipAdapter", 2), seed=7)
    model.clip_adapter = AdaptedClipAdapter("tiny", 2, True, arch=arch, precision="fp32")
    model.load_state_dict(sd)
    K = 7
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("adapted_val").set(thing_classes=names)
    g = torch.Generator().manual_seed(1)
    base = torch.randn(1, 64, generator=g)
    text = F.normalize(base + 0.05 * torch.randn(K, 64, generator=g), dim=-1)
    model.clip_adapter.set_text_features(names, text)
    frames = (torch.rand(2, 3, 90, 120, generator=g) * 255).to(torch.uint8)
    st, ref_st = {}, {}
    model([{"image": [f for f in frames], "dataset_name": "adapted_val"}], stages=st)
    with torch.no_grad():
        TR.openvis_forward(frames, sd, text, stages=ref_st, clip_heads=4, clip_resolution=64, mask_prompt_depth=2)
        ref_plain = {}
        TR.openvis_forward(frames, sd, text, stages=ref_plain, clip_heads=4, clip_resolution=64)
    vg, vr = st["valid"], ref_st["valid"].numpy()
    ig = {tuple(x): i for i, x in enumerate(np.argwhere(vg))}
    ir = {tuple(x): i for i, x in enumerate(np.argwhere(vr))}
    common = [k for k in ig if k in ir]
    assert len(common) > 0.95 * len(ir)
    lg, lr, lp = st["crop_logits"].cpu().numpy(), ref_st["crop_logits"].numpy(), ref_plain["crop_logits"].numpy()
    d = np.array([np.abs(lg[ig[k]] - lr[ir[k]]).max() for k in common])
    assert np.median(d) < 1e-2 and (d < 1e-1).mean() > 0.97, (np.median(d), (d < 1e-1).mean())
    assert np.abs(lr - lp).max() > 0.5                                    # the prompt changes the logits on this clip


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("fp16", 2e-3)])
@pytest.mark.parametrize("B", [1, 3, 70])
def test_last_block_for_the_class_token_equals_the_full_block(precision, tol, B):
    """ClipVisual.last_block_cls (queries / out-proj / MLP for token 0 only) against run_blocks on every token"""
    from openvis_amd import weights
    from openvis_amd.modeling.clip_adapter.adapter import ClipVisual
    arch = dict(width=256, layers=3, heads=4, patch=16, resolution=64, embed_dim=64)
    prefix = "clip_adapter.clip_model.visual."
    sd = weights.random_init(weights.clip_visual_spec(**arch), seed=41)
    vis = ClipVisual(**arch, precision=precision).load_state_dict(sd, prefix, "cuda")
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 17, 256, generator=g).cuda()
    full = vis.run_blocks(x.clone(), 2, 3)[:, 0, :]
    cls = vis.last_block_cls(x.clone(), 2)
    assert cls.shape == (B, 256)
    assert (cls - full).abs().max().item() / full.abs().max().item() < tol


@pytest.mark.parametrize("arch", [dict(width=768, layers=3, heads=12, patch=16, resolution=224, embed_dim=512),
                                  dict(width=1024, layers=2, heads=16, patch=14, resolution=336, embed_dim=768)])
def test_layernorm_folded_tower_agrees_with_the_layernorm_kernels(arch):
    """fp16 residual stream at production width: ln_1 / ln_2 folded into in_proj / c_fc (ViT-B: both; ViT-L: in_proj only, c_fc's 4096 columns
    do not fit) and the statistics taken from the epilogue of out_proj / c_proj, against LayerNorm kernel + plain GEMM on the same weights
    with non-trivial gamma / beta; and both against the f32-policy tower."""
    from openvis_amd import ops, weights
    from openvis_amd.modeling.clip_adapter.adapter import ClipVisual
    prefix = "clip_adapter.clip_model.visual."
    sd = weights.random_init(weights.clip_visual_spec(**arch), seed=43)
    g = torch.Generator().manual_seed(1)
    for k in list(sd):
        if ".ln_1." in k or ".ln_2." in k:
            sd[k] = sd[k] + (0.3 if k.endswith("weight") else 0.2) * torch.randn(sd[k].shape, generator=g)
    vis = ClipVisual(**arch, precision="fp16").load_state_dict(sd, prefix, "cuda")
    vis32 = ClipVisual(**arch, precision="fp32").load_state_dict(sd, prefix, "cuda")
    L = (arch["resolution"] // arch["patch"]) ** 2 + 1
    B = 360 if arch["width"] == 768 else 120                       # >= 256 tiles of 256 x 256 for every GEMM of the block
    x = torch.randn(B, L, arch["width"], generator=g).cuda()
    x16 = ops.cast_f16(x)
    n = arch["layers"]
    vis.fold_ln = True
    calls = []
    real_ln = ops.gemm_nt_f16_ln
    ops.gemm_nt_f16_ln = lambda *a, **k: (calls.append(a[1].shape[0]), real_ln(*a, **k))[1]
    try:
        y_fold, st_fold = vis.run_blocks(x16.clone(), 0, n - 1, with_stats=True)     # explicit hand-over of the GEMM's row statistics
        c_fold = vis.last_block_cls(y_fold, n - 1, st_fold)
    finally:
        ops.gemm_nt_f16_ln = real_ln
    C = arch["width"]
    assert calls.count(3 * C) == n - 1 and calls.count(2 * C) == 1                  # in_proj of every block, keys | values of the last
    assert calls.count(4 * C) == (n - 1 if C == 768 else 0)                        # c_fc: ViT-B only
    vis.fold_ln = False
    y_plain = vis.run_blocks(x16.clone(), 0, n - 1)
    c_plain = vis.last_block_cls(y_plain, n - 1)
    y32 = vis32.run_blocks(x16.float(), 0, n - 1)
    c32 = vis32.last_block_cls(y32, n - 1)
    scale = c32.abs().max().item()
    d_fold, d_plain = (c_fold - c32).abs().max().item() / scale, (c_plain - c32).abs().max().item() / scale
    assert d_fold < 4e-3 and d_plain < 4e-3, (d_fold, d_plain)
    assert (c_fold - c32).abs().mean().item() <= 1.1 * (c_plain - c32).abs().mean().item()      # the fold skips one fp16 rounding: not worse
    assert (y_fold.float() - y_plain.float()).abs().max().item() / y32.abs().max().item() < 4e-3


def test_second_load_state_dict_replaces_the_folded_operands():
    """ClipVisual caches LayerNorm-folded operands (ops.fold_layernorm) at first use.  A second load_state_dict on the same adapter must not
    meet the first checkpoint's folds (round 6: it did -- the cache lived in the weight dict, which load_state_dict kept): after reloading,
    the adapter equals a freshly built one on the new weights, bit for bit."""
    import bench
    from openvis_amd import weights
    from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter
    arch = dict(width=256, layers=2, heads=4, patch=16, resolution=64, embed_dim=64)
    spec = [(k, s) for k, s in weights.openvis_spec("r50", arch, 100) if k.startswith("clip_adapter.")]
    sd_a, sd_b = weights.random_init(spec, seed=1), weights.random_init(spec, seed=2)
    g = torch.Generator().manual_seed(3)
    A = torch.randn(6 * 16, 768, generator=g).half().cuda()
    outs = []
    for first in (sd_a, None):
        ad = ClipAdapter("tiny", arch=arch, precision="fp16")
        ad.visual.stream16 = True
        if first is not None:
            ad.load_state_dict(first)
            ad.visual.forward_patches(A, 6)                         # builds the folds of checkpoint A
        ad.load_state_dict(sd_b)
        outs.append(ad.visual.forward_patches(A, 6).float().cpu())
    assert torch.equal(outs[0], outs[1])
