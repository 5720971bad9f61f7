"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz by importing the REFERENCE's own
Python modules from /root/reference (build container only; see oracle/ref_import.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [name ...]
The committed fixtures are data (inputs + the reference's outputs); no reference source travels.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

import numpy as np
import torch

from oracle import ref_import as R


def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def gen_msda():
    f = R.ref("openvis.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    core = f.ms_deform_attn_core_pytorch
    out = {}
    # --- the reference's own fixture: ops/test.py:24-39 (seed 3; double check first, then float) ---
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = _lsi(shapes)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for tag in ("double", "float"):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            o = core(value.double(), shapes, loc.double(), w.double())
        else:
            o = core(value, shapes, loc, w)
        out[f"testpy_{tag}_value"] = value.numpy()
        out[f"testpy_{tag}_loc"] = loc.numpy()
        out[f"testpy_{tag}_w"] = w.numpy()
        out[f"testpy_{tag}_out"] = o.numpy()
    out["testpy_shapes"] = shapes.numpy()
    out["testpy_lsi"] = lsi.numpy()

    # --- model-shaped and edge cases (locations reach outside [0,1] to hit the border rules) ---
    cases = {
        "enc": dict(N=2, M=8, D=32, L=3, P=4, shapes=[(2, 3), (4, 6), (8, 12)], Lq=None),
        "oddD": dict(N=1, M=3, D=30, L=2, P=3, shapes=[(5, 3), (2, 9)], Lq=7),
        "wide": dict(N=1, M=2, D=71, L=1, P=5, shapes=[(3, 4)], Lq=5),
        "L4": dict(N=3, M=4, D=16, L=4, P=4, shapes=[(1, 1), (2, 3), (6, 5), (9, 2)], Lq=11),
    }
    g = torch.Generator().manual_seed(1234)
    for name, c in cases.items():
        shapes = torch.as_tensor(c["shapes"], dtype=torch.long)
        lsi = _lsi(shapes)
        S = int(shapes.prod(1).sum())
        Lq = c["Lq"] or S
        value = torch.randn(c["N"], S, c["M"], c["D"], generator=g)
        loc = torch.rand(c["N"], Lq, c["M"], c["L"], c["P"], 2, generator=g) * 1.5 - 0.25
        w = torch.softmax(torch.randn(c["N"], Lq, c["M"], c["L"] * c["P"], generator=g), -1).view(
            c["N"], Lq, c["M"], c["L"], c["P"])
        o32 = core(value, shapes, loc, w)
        o64 = core(value.double(), shapes, loc.double(), w.double())
        out[f"{name}_value"] = value.numpy()
        out[f"{name}_loc"] = loc.numpy()
        out[f"{name}_w"] = w.numpy()
        out[f"{name}_shapes"] = shapes.numpy()
        out[f"{name}_lsi"] = lsi.numpy()
        out[f"{name}_out32"] = o32.numpy()
        out[f"{name}_out64"] = o64.numpy()
    np.savez_compressed(os.path.join(GOLD, "msda.npz"), **out)
    print("wrote msda.npz", {k: v.shape for k, v in out.items() if k.endswith("out32") or k.endswith("_out")})


def _spec_arrays(spec):
    import json
    return np.frombuffer(json.dumps([[k, list(s)] for k, s in spec]).encode(), dtype=np.uint8)


def _build_pixel_decoder():
    md = R.ref("openvis.modeling.pixel_decoder.msdeformattn")
    SS = sys.modules["detectron2.layers"].ShapeSpec
    inshape = {"res2": SS(channels=256, stride=4), "res3": SS(channels=512, stride=8),
               "res4": SS(channels=1024, stride=16), "res5": SS(channels=2048, stride=32)}
    return md.MSDeformAttnPixelDecoder(
        inshape, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
        transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()


def _load_synth(module, seed):
    from tests._synth import synth_weights, spec_of
    spec = spec_of(module.state_dict())
    missing = module.load_state_dict(synth_weights(spec, seed), strict=False)
    assert not missing.unexpected_keys and all(not module.state_dict()[k].dtype.is_floating_point for k in missing.missing_keys), missing
    return spec            # integer buffers (e.g. Swin's relative_position_index) keep their constructed values


PD_SEED, PD_IN_SEED, DEC_SEED, CLIP_SEED = 101, 102, 103, 104
PD_T, PD_H, PD_W = 2, 32, 64


def pd_input_shapes(T=PD_T, H=PD_H, W=PD_W):
    return [(T, 256, H // 4, W // 4), (T, 512, H // 8, W // 8), (T, 1024, H // 16, W // 16), (T, 2048, H // 32, W // 32)]


def gen_pixel_decoder_and_decoder():
    """Reference MSDeformAttnPixelDecoder.forward_features (msdeformattn.py:329-380) and
    VideoMultiScaleMaskedTransformerDecoder.forward (video decoder:380-452) on synthetic weights."""
    from tests._synth import synth_inputs
    pd = _build_pixel_decoder()
    spec_pd = _load_synth(pd, PD_SEED)
    f = synth_inputs(pd_input_shapes(), PD_IN_SEED)
    feats = dict(zip(["res2", "res3", "res4", "res5"], f))
    vd = R.ref("openvis.modeling.transformer_decoder.video_mask2former_transformer_decoder")
    dec = vd.VideoMultiScaleMaskedTransformerDecoder(
        256, True, num_classes=1, hidden_dim=256, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9,
        pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=2).eval()
    spec_dec = _load_synth(dec, DEC_SEED)
    with torch.no_grad():
        mf, o0, ms = pd.forward_features(feats)
        out = dec(ms, mf)
    d = dict(spec_pd=_spec_arrays(spec_pd), spec_dec=_spec_arrays(spec_dec),
             seeds=np.array([PD_SEED, PD_IN_SEED, DEC_SEED]), thw=np.array([PD_T, PD_H, PD_W]),
             mask_features=mf.numpy(), ms0=ms[0].numpy(), ms1=ms[1].numpy(), ms2=ms[2].numpy(),
             pred_logits=out["pred_logits"].numpy(), pred_masks=out["pred_masks"].numpy(),
             aux_masks_0=out["aux_outputs"][0]["pred_masks"].numpy(),
             aux_masks_5=out["aux_outputs"][5]["pred_masks"].numpy())
    np.savez_compressed(os.path.join(GOLD, "pixel_decoder_decoder.npz"), **d)
    print("wrote pixel_decoder_decoder.npz", mf.shape, out["pred_masks"].shape)


def gen_clip_visual():
    """Reference VisionTransformer.forward(x, m=None) (mask_adapted_clip/model.py:327-362) on a tiny ViT."""
    from tests._synth import synth_inputs
    m = R.ref("mask_adapted_clip.model")
    vit = m.VisionTransformer(input_resolution=64, patch_size=16, mask_prompt_depth=0, width=128, layers=3,
                              heads=2, output_dim=64).eval()
    spec = _load_synth(vit, CLIP_SEED)
    x = synth_inputs([(3, 3, 64, 64)], CLIP_SEED + 1)[0]
    with torch.no_grad():
        y = vit(x)
    np.savez_compressed(os.path.join(GOLD, "clip_visual_tiny.npz"), spec=_spec_arrays(spec),
                        seeds=np.array([CLIP_SEED, CLIP_SEED + 1]), out=y.numpy())
    print("wrote clip_visual_tiny.npz", y.shape)


def gen_clip_visual_mask_prompt():
    """Reference VisionTransformer.forward(x, m) with mask_prompt_depth=3 (mask_adapted_clip/model.py:327-362): the
    mask-prompt path AdaptedClipAdapter.encode_image drives (mask_adapted_adapter.py:143-147)."""
    from tests._synth import synth_inputs
    mod = R.ref("mask_adapted_clip.model")
    vit = mod.VisionTransformer(input_resolution=64, patch_size=16, mask_prompt_depth=3, width=256, layers=4,
                                heads=4, output_dim=64).eval()
    spec = _load_synth(vit, CLIP_SEED + 10)
    x = synth_inputs([(4, 3, 64, 64)], CLIP_SEED + 11)[0]
    # mask regions [M,1,64,64] >= 0 with whole patches at exactly 0 (what roi_align gives outside the image) and one
    # patch that is 0 except for a single tiny positive value (ceil of the pooled mean must still open it)
    m = synth_inputs([(4, 1, 64, 64)], CLIP_SEED + 12)[0].abs()
    m[0, 0, :16, :] = 0
    m[1, 0, :, 48:] = 0
    m[2, 0, 16:48, 16:48] = 0
    m[2, 0, 20, 20] = 1e-6
    m[3] = 0
    m[3, 0, 33, 1] = 0.25
    with torch.no_grad():
        y = vit(x, m)
    np.savez_compressed(os.path.join(GOLD, "clip_visual_mask_prompt.npz"), spec=_spec_arrays(spec),
                        seeds=np.array([CLIP_SEED + 10, CLIP_SEED + 11, CLIP_SEED + 12]), mask=m.numpy(), out=y.numpy())
    print("wrote clip_visual_mask_prompt.npz", y.shape)


def gen_position_encodings():
    pe2 = R.ref("openvis.modeling.pixel_decoder.position_encoding").PositionEmbeddingSine2D(128, normalize=True)
    pe3 = R.ref("openvis.modeling.transformer_decoder.position_encoding").PositionEmbeddingSine3D(128, normalize=True)
    a = pe2(torch.zeros(1, 1, 5, 7))
    b = pe3(torch.zeros(1, 3, 1, 4, 6))
    np.savez_compressed(os.path.join(GOLD, "position_encodings.npz"), pe2d_5x7=a.numpy(), pe3d_3x4x6=b.numpy())
    print("wrote position_encodings.npz")


def gen_frame_decoder_and_tracker():
    """Reference FrameMultiScaleMaskedTransformerDecoder.forward (frame decoder:52-137) and the MinVIS tracker
    (minvis.py:28-72) on synthetic weights / inputs."""
    from tests._synth import synth_inputs
    fd = R.ref("openvis.modeling.transformer_decoder.frame_mask2former_transformer_decoder")
    mv = R.ref("openvis.modeling.minvis")
    T = 3
    dec = fd.FrameMultiScaleMaskedTransformerDecoder(
        256, True, num_classes=1, hidden_dim=256, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9,
        pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T).eval()
    spec = _load_synth(dec, 111)
    # Random-init mask logits sit near 0, so a 1e-5 rounding difference between two correct implementations can flip
    # one attention-mask bit and visibly change a whole frame downstream (seen with input seed 112, frame 0, layer 9).
    # Pick the first input seed whose run has a clear margin: no level-resolution mask logit within 1e-3 of 0.
    from oracle import torch_ref as TR
    from tests._synth import synth_weights
    Wd = synth_weights(spec, 111, "sem_seg_head.predictor.")
    for s_ms in range(112, 200, 10):
        ms = synth_inputs([(T, 256, 2, 3), (T, 256, 4, 6), (T, 256, 8, 12)], s_ms)
        mf = synth_inputs([(T, 256, 16, 24)], s_ms + 1)[0]
        with torch.no_grad():
            out = dec(ms, mf)
            mine = TR.frame_decoder(ms, mf, Wd)
        if (out["pred_masks"] - mine["pred_masks"]).abs().max() < 1e-3:
            break
    else:
        raise RuntimeError("no stable seed found")
    print("frame decoder fixture uses input seed", s_ms)
    with torch.no_grad():
        idx, emb = mv.batch_video_match_via_embeds(out["pred_embeds"])
    # a harder tracker case: noisy permutations of a base embedding set
    g = torch.Generator().manual_seed(114)
    base = torch.randn(40, 64, generator=g)
    seq = torch.stack([base[torch.randperm(40, generator=g)] + 0.3 * torch.randn(40, 64, generator=g) for _ in range(6)])
    idx2, _ = mv.batch_video_match_via_embeds(seq[None])
    np.savez_compressed(os.path.join(GOLD, "frame_decoder_tracker.npz"), spec=_spec_arrays(spec),
                        seeds=np.array([111, s_ms, s_ms + 1, 114]), pred_logits=out["pred_logits"].numpy(),
                        pred_masks=out["pred_masks"].numpy(), pred_embeds=out["pred_embeds"].numpy(),
                        indices=idx.numpy(), track_seq=seq.numpy(), track_indices=idx2.numpy())
    print("wrote frame_decoder_tracker.npz", out["pred_masks"].shape, idx.shape)


def gen_decoder_head_variants():
    """Reference Embedding* / Proposal* decoder variants (frame decoder:157-207, video decoder:487-537): the parent decoder with another
    class head.  They get the PARENT fixture's weights (same seeds, so the stable inputs of those fixtures stay stable) plus separately
    seeded class_embed weights; stored: pred_logits of each variant (the masks are checked here to equal the parent fixture's bit for bit and are not stored again)."""
    import json
    from tests._synth import synth_inputs, synth_weights, spec_of
    fd = R.ref("openvis.modeling.transformer_decoder.frame_mask2former_transformer_decoder")
    vd = R.ref("openvis.modeling.transformer_decoder.video_mask2former_transformer_decoder")
    gf = np.load(os.path.join(GOLD, "frame_decoder_tracker.npz"))
    gv = np.load(os.path.join(GOLD, "pixel_decoder_decoder.npz"))
    spec_of_arr = lambda arr: [(k, tuple(sh)) for k, sh in json.loads(bytes(arr.tolist()).decode())]
    s_dec, s_ms, s_mf, _ = [int(x) for x in gf["seeds"]]
    T = 3
    kw = dict(num_classes=1, hidden_dim=256, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256,
              enforce_input_project=False)
    ms_f = synth_inputs([(T, 256, 2, 3), (T, 256, 4, 6), (T, 256, 8, 12)], s_ms)
    mf_f = synth_inputs([(T, 256, 16, 24)], s_mf)[0]
    ms_v = [torch.from_numpy(gv[f"ms{i}"]) for i in range(3)]
    mf_v = torch.from_numpy(gv["mask_features"])
    out = {}
    HEAD_SEED, CLIP_DIMS = 181, 64
    cases = {"embedding_frame": (fd.EmbeddingFrameMultiScaleMaskedTransformerDecoder, dict(clip_dims=CLIP_DIMS), spec_of_arr(gf["spec"]), s_dec, T, ms_f, mf_f),
             "proposal_frame": (fd.ProposalFrameMultiScaleMaskedTransformerDecoder, {}, spec_of_arr(gf["spec"]), s_dec, T, ms_f, mf_f),
             "embedding_video": (vd.EmbeddingVideoMultiScaleMaskedTransformerDecoder, dict(clip_dims=CLIP_DIMS), spec_of_arr(gv["spec_dec"]), int(gv["seeds"][2]), 2, ms_v, mf_v),
             "proposal_video": (vd.ProposalVideoMultiScaleMaskedTransformerDecoder, {}, spec_of_arr(gv["spec_dec"]), int(gv["seeds"][2]), 2, ms_v, mf_v)}
    for name, (cls, extra, parent_spec, parent_seed, nf, ms, mf) in cases.items():
        dec = cls(mask_classification=True, in_channels=256, num_frames=nf, **extra, **kw).eval()
        parent = {k: v for k, v in synth_weights(parent_spec, parent_seed).items() if not k.startswith("class_embed")}
        head_spec = [(k, sh) for k, sh in spec_of(dec.state_dict()) if k.startswith("class_embed")]
        head = synth_weights(head_spec, HEAD_SEED)
        missing = dec.load_state_dict({**parent, **head}, strict=False)
        assert not missing.unexpected_keys and not missing.missing_keys, missing
        with torch.no_grad():
            o = dec(ms, mf)
        out[name + "_logits"] = o["pred_logits"].numpy()
        assert np.array_equal(o["pred_masks"].numpy(), (gf if "frame" in name else gv)["pred_masks"])     # the parent fixture's masks, bit for bit
        out[name + "_head_spec"] = _spec_arrays(head_spec)
        print(name, o["pred_logits"].shape, o["pred_masks"].shape)
    np.savez_compressed(os.path.join(GOLD, "decoder_head_variants.npz"), head_seed=np.array([HEAD_SEED]), clip_dims=np.array([CLIP_DIMS]), **out)
    print("wrote decoder_head_variants.npz")


SAN_CLIP = dict(embed_dim=64, image_resolution=64, vision_layers=4, vision_width=256, vision_patch_size=16,
                mask_prompt_depth=0, context_length=8, vocab_size=64, transformer_width=64, transformer_heads=2,
                transformer_layers=1)


def gen_side_adapter():
    """Reference SideAdapter.front_encode_image / post_encode_image / cal_sim_logits (side_adapter.py:147-235) and
    SideAdapterFrameMultiScaleMaskedTransformerDecoder.forward (side-frame decoder:57-169) on a tiny CLIP."""
    from tests._synth import synth_inputs, synth_weights, spec_of
    from oracle import torch_ref as TR
    sa = R.ref("openvis.modeling.clip_adapter.side_adapter")
    mac = R.ref("mask_adapted_clip.model")
    sa.build_clip_model = lambda name: mac.CLIP(**SAN_CLIP)          # clip.load() needs the network
    Q, T = 12, 2
    ad = sa.SideAdapter("tiny", out_dims=256, broken_idx=3, merge_ids=[1, 2, 3], num_queries=Q, text_templates=["{}"]).eval()
    spec_ad = _load_synth(ad, 131)
    frames = (synth_inputs([(T, 3, 96, 128)], 132)[0].sigmoid() * 255).floor()      # raw padded frames 0..255
    sfd = R.ref("openvis.modeling.transformer_decoder.side_adapter_frame_mask2former_transformer_decoder")
    dec = sfd.SideAdapterFrameMultiScaleMaskedTransformerDecoder(
        clip_heads=4, mask_classification=True, in_channels=256, num_classes=1, hidden_dim=256, num_queries=Q, nheads=8,
        dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T).eval()
    spec_dec = _load_synth(dec, 133)
    Wd = synth_weights(spec_dec, 133, "sem_seg_head.predictor.")
    for s_ms in range(134, 234, 10):                                   # stable seed (see gen_frame_decoder_and_tracker)
        ms = synth_inputs([(T, 256, 3, 4), (T, 256, 6, 8), (T, 256, 12, 16)], s_ms)
        mf = synth_inputs([(T, 256, 24, 32)], s_ms + 1)[0]
        with torch.no_grad():
            out = dec(ms, mf)
            mine = TR.side_frame_decoder(ms, mf, Wd, clip_heads=4)
        if (out["pred_masks"] - mine["pred_masks"]).abs().max() < 1e-3:
            break
    else:
        raise RuntimeError("no stable seed found")
    print("side-frame decoder fixture uses input seed", s_ms)
    text = torch.nn.functional.normalize(synth_inputs([(5, 64)], 136)[0], dim=-1)
    with torch.no_grad():
        mg, bk = ad.front_encode_image(frames)
        biases = out["class_attn_biases"].flatten(0, 1)                # [T, n, Q, ha, wa]
        sos = ad.post_encode_image(bk, biases)
        tf = torch.cat([text, torch.nn.functional.normalize(ad.bg_embed, dim=-1)], dim=0)
        logits = ad.cal_sim_logits(tf, sos)
    np.savez_compressed(os.path.join(GOLD, "side_adapter.npz"), spec_ad=_spec_arrays(spec_ad), spec_dec=_spec_arrays(spec_dec),
                        seeds=np.array([131, 132, 133, s_ms, s_ms + 1, 136]), mg0=mg[0].numpy(), mg1=mg[1].numpy(),
                        mg2=mg[2].numpy(), bk_cls=bk[0].numpy(), bk_pix=bk[1].numpy(), sos=sos.numpy(),
                        logits=logits.numpy(), class_attn_biases=out["class_attn_biases"].numpy(),
                        pred_masks=out["pred_masks"].numpy(), pred_embeds=out["pred_embeds"].numpy())
    print("wrote side_adapter.npz", sos.shape, logits.shape, out["class_attn_biases"].shape)


def gen_side_video_decoder():
    """Reference SideAdapterVideoMultiScaleMaskedTransformerDecoder.forward (side-video decoder:51-142), eval: one query
    set for the whole clip, per-frame per-head attention biases."""
    from tests._synth import synth_inputs, synth_weights
    from oracle import torch_ref as TR
    svd = R.ref("openvis.modeling.transformer_decoder.side_adapter_video_mask2former_transformer_decoder")
    Q, T = 12, 3
    dec = svd.SideAdapterVideoMultiScaleMaskedTransformerDecoder(
        clip_heads=4, mask_classification=True, in_channels=256, num_classes=1, hidden_dim=256, num_queries=Q, nheads=8,
        dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T).eval()
    spec = _load_synth(dec, 171)
    Wd = synth_weights(spec, 171, "sem_seg_head.predictor.")
    for s_ms in range(172, 272, 10):                                   # stable seed (see gen_frame_decoder_and_tracker)
        ms = synth_inputs([(T, 256, 3, 4), (T, 256, 6, 8), (T, 256, 12, 16)], s_ms)
        mf = synth_inputs([(T, 256, 24, 32)], s_ms + 1)[0]
        with torch.no_grad():
            out = dec(ms, mf)
            mine = TR.side_video_decoder(ms, mf, Wd, clip_heads=4)
        if (out["pred_masks"] - mine["pred_masks"]).abs().max() < 1e-3:
            break
    else:
        raise RuntimeError("no stable seed found")
    print("side-video decoder fixture uses input seed", s_ms, "oracle diff",
          (out["pred_masks"] - mine["pred_masks"]).abs().max().item(), (out["class_attn_biases"] - mine["class_attn_biases"]).abs().max().item())
    np.savez_compressed(os.path.join(GOLD, "side_video_decoder.npz"), spec=_spec_arrays(spec), seeds=np.array([171, s_ms, s_ms + 1]),
                        dims=np.array([Q, T]), class_attn_biases=out["class_attn_biases"].numpy(), pred_masks=out["pred_masks"].numpy())
    print("wrote side_video_decoder.npz", out["class_attn_biases"].shape, out["pred_masks"].shape)


def gen_resampler():
    """Reference TemporalInstanceResampler.forward (resampler.py:244-316) with a stub adapter (the CLIP pass of its
    prediction heads is covered by the side_adapter fixture)."""
    from tests._synth import synth_inputs
    rs = R.ref("openvis.modeling.resampler")
    m = rs.TemporalInstanceResampler(hidden_dim=256, feed_dim=2048, nheads=8, nlayers=6).eval()
    spec = _load_synth(m, 141)
    T, Q, n = 5, 9, 2
    fe = synth_inputs([(1, T, Q, 256)], 142)[0]
    mf = synth_inputs([(T, 256, 8, 12)], 143)[0]
    af = synth_inputs([(T, n, 256, 2, 3)], 144)[0]

    class _Adapter:                                   # records the biases, returns them pooled as fake "logits"
        def post_encode_image(self, bk, biases):
            self.biases = biases
            return biases.mean(dim=(1, 3, 4))[..., None]
        def cal_sim_logits(self, text, feats):
            return feats
    ad = _Adapter()
    with torch.no_grad():
        out = m(fe, mf, af, ad, None, None)
    np.savez_compressed(os.path.join(GOLD, "resampler.npz"), spec=_spec_arrays(spec), seeds=np.array([141, 142, 143, 144]),
                        dims=np.array([T, Q, n]), pred_masks=out["pred_masks"].numpy(), pred_embeds=out["pred_embeds"].numpy(),
                        last_biases=ad.biases.numpy())
    print("wrote resampler.npz", out["pred_masks"].shape, ad.biases.shape)


def gen_swin():
    """Reference SwinTransformer.forward (backbone/swin.py:702-722): head_dim 32, window 5 (so every stage pads), both
    shifted and unshifted blocks, odd-sized patch merging."""
    from tests._synth import synth_inputs
    sw = R.ref("openvis.modeling.backbone.swin")
    cfg = dict(embed_dim=64, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], window_size=5)
    m = sw.SwinTransformer(pretrain_img_size=224, patch_size=4, in_chans=3, drop_path_rate=0.0, **cfg)
    m.eval()                                           # the reference's train() override returns None
    spec = _load_synth(m, 151)
    x = synth_inputs([(2, 3, 64, 104)], 152)[0]
    with torch.no_grad():
        out = m(x)
    np.savez_compressed(os.path.join(GOLD, "swin.npz"), spec=_spec_arrays(spec), seeds=np.array([151, 152]),
                        cfg=np.array([cfg["embed_dim"], cfg["window_size"]] + cfg["depths"] + cfg["num_heads"]),
                        **{k: v.numpy() for k, v in out.items()})
    print("wrote swin.npz", {k: tuple(v.shape) for k, v in out.items()})


def gen_swin_ape():
    """Reference SwinTransformer.forward with ape=True (backbone/swin.py:567-578, 706-713): the absolute position embedding of
    the 56x56 pretrain grid (pretrain_img_size 224) is interpolated bicubically to the 16x26 patch grid of the input."""
    from tests._synth import synth_inputs
    sw = R.ref("openvis.modeling.backbone.swin")
    cfg = dict(embed_dim=32, depths=[1, 1, 1, 1], num_heads=[1, 2, 4, 8], window_size=5)
    m = sw.SwinTransformer(pretrain_img_size=224, patch_size=4, in_chans=3, drop_path_rate=0.0, ape=True, **cfg)
    m.eval()
    spec = _load_synth(m, 161)
    x = synth_inputs([(2, 3, 64, 104)], 162)[0]
    with torch.no_grad():
        out = m(x)
    np.savez_compressed(os.path.join(GOLD, "swin_ape.npz"), spec=_spec_arrays(spec), seeds=np.array([161, 162]),
                        cfg=np.array([cfg["embed_dim"], cfg["window_size"]] + cfg["depths"] + cfg["num_heads"]),
                        **{k: v.numpy() for k, v in out.items()})
    print("wrote swin_ape.npz", {k: tuple(v.shape) for k, v in out.items()})


TEXT_NOUNS = ["person", "traffic light", "hot dog", "teddy bear", "skateboard", "giant_panda", "earless_seal", "ape"]


def gen_clip_text():
    """Reference tokenizer (mask_adapted_clip/simple_tokenizer.py + clip.py:239-283 tokenize) on the 14 "vild" templates x a
    few class names, and the reference CLIP.encode_text + ClipAdapter.encode_text prompt ensemble (adapter.py:121-138) on a
    tiny text tower (width 64, 1 head of 64, 2 layers, embed 32; the 49 408-entry vocabulary is the real one)."""
    import types
    R.install()
    sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda t: t))        # ftfy is absent; ASCII input: identity
    tk = R.ref("mask_adapted_clip.simple_tokenizer").SimpleTokenizer()
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    templates = R.ref("openvis.modeling.clip_adapter.text_prompt").PREDEFINED_TEMPLATES["vild"]

    def tokenize(text):                                                                # clip.py:266-283
        ids = [sot] + tk.encode(text) + [eot]
        out = np.zeros(77, np.int64)
        out[:len(ids)] = ids
        return out
    tokens = np.stack([np.stack([tokenize(t.format(n)) for n in TEXT_NOUNS]) for t in templates])      # [14, K, 77]
    odd = ["Hello, World!!  it's  a café's dog", "3 cats & 12dogs (x-ray)", "  trailing   spaces ", "naïve façade", "a_b-c/d"]
    odd_tok = np.stack([tokenize(t) for t in odd])
    mm = R.ref("mask_adapted_clip.model")
    clip = mm.CLIP(embed_dim=32, image_resolution=32, vision_layers=1, vision_width=64, vision_patch_size=16, context_length=77,
                   vocab_size=49408, transformer_width=64, transformer_heads=1, transformer_layers=2, mask_prompt_depth=0).eval()
    spec = [(k, s) for k, s in __import__("tests._synth", fromlist=["spec_of"]).spec_of(clip.state_dict())
            if not k.startswith("visual.")]
    from tests._synth import synth_weights
    sd = synth_weights(spec, 161)
    clip.load_state_dict(sd, strict=False)
    with torch.no_grad():
        bucket = []
        for t in range(tokens.shape[0]):
            e = clip.encode_text(torch.from_numpy(tokens[t]))
            if t == 0:
                raw0 = e.clone()
            bucket.append(e / e.norm(dim=-1, keepdim=True))
        ens = torch.stack(bucket).mean(dim=0)
        ens = ens / ens.norm(dim=-1, keepdim=True)
    np.savez_compressed(os.path.join(GOLD, "clip_text.npz"), spec=_spec_arrays(spec), seeds=np.array([161]),
                        tokens=tokens.astype(np.int32), odd_tokens=odd_tok.astype(np.int32),
                        odd_texts=np.frombuffer("\n".join(odd).encode("utf-8"), dtype=np.uint8),
                        nouns=np.frombuffer("\n".join(TEXT_NOUNS).encode(), dtype=np.uint8),
                        encode_text_template0=raw0.numpy(), ensemble=ens.numpy())
    print("wrote clip_text.npz", tokens.shape, ens.shape)


def gen_window_inference():
    """Reference MinVIS.run_window_inference (minvis.py:340-362) + MinVIS.post_processing (:320-338) and SAN.run_window_inference
    (SANOnline, san.py:285-307), called as UNBOUND methods on a stub `self` whose `backbone` is the oracle's ResNet-50 restatement (detectron2's
    is not in /root/reference) and whose `sem_seg_head` is the reference's own pixel decoder + frame decoder (MinVIS) or pixel decoder
    with extra features + side-adapter frame decoder (SAN).  5 frames of 64 x 96 in windows of 2 (2 + 2 + 1)."""
    import types
    from tests._synth import synth_inputs, synth_weights
    from oracle import torch_ref as TR
    from openvis_amd import weights as PW                                      # key / shape list of the R50 backbone only
    mv = R.ref("openvis.modeling.minvis")
    fd = R.ref("openvis.modeling.transformer_decoder.frame_mask2former_transformer_decoder")
    sfd = R.ref("openvis.modeling.transformer_decoder.side_adapter_frame_mask2former_transformer_decoder")
    sa = R.ref("openvis.modeling.clip_adapter.side_adapter")
    sys.modules["openvis.modeling.clip_adapter"].SideAdapter = sa.SideAdapter
    R.ref("openvis.modeling.clip_adapter.text_prompt")
    san = R.ref("openvis.san")
    T, H, Wd, WIN, Q = 5, 64, 96, 2, 100
    tiny = dict(width=64, layers=1, heads=1, patch=16, resolution=32, embed_dim=16)
    spec_bb = [(k[len("backbone."):], s) for k, s in PW.openvis_spec("r50", tiny, Q) if k.startswith("backbone.")]
    Wbb = synth_weights(spec_bb, 171, "backbone.")
    pd = _build_pixel_decoder()
    spec_pd = _load_synth(pd, 172)
    dec = fd.FrameMultiScaleMaskedTransformerDecoder(
        256, True, num_classes=1, hidden_dim=256, num_queries=Q, nheads=8, dim_feedforward=2048, dec_layers=9,
        pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T).eval()
    spec_dec = _load_synth(dec, 173)
    sdec = sfd.SideAdapterFrameMultiScaleMaskedTransformerDecoder(
        clip_heads=4, mask_classification=True, in_channels=256, num_classes=1, hidden_dim=256, num_queries=Q, nheads=8,
        dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T).eval()
    spec_sdec = _load_synth(sdec, 174)
    Wall = dict(Wbb)
    Wall.update(synth_weights(spec_pd, 172, "sem_seg_head.pixel_decoder."))
    Wall.update(synth_weights(spec_dec, 173, "sem_seg_head.predictor."))
    Wsan = synth_weights(spec_sdec, 174, "sem_seg_head.predictor.")

    def backbone(x):
        with torch.no_grad():
            return TR.resnet50(x, Wbb)

    def head(features, extra_feats=None):
        with torch.no_grad():
            mf, _, ms = pd.forward_features(features, extra_feats) if extra_feats is not None else pd.forward_features(features)
            return (sdec if extra_feats is not None else dec)(ms, mf)

    stub = types.SimpleNamespace(backbone=backbone, sem_seg_head=head)
    for s_in in range(175, 275, 10):                                        # stable seed (see gen_frame_decoder_and_tracker)
        frames = (synth_inputs([(T, 3, H, Wd)], s_in)[0].sigmoid() * 255).floor().to(torch.uint8)
        images, _ = TR.preprocess([f for f in frames])
        mg = synth_inputs([(T, 256, H // 32, Wd // 32), (T, 256, H // 16, Wd // 16), (T, 256, H // 8, Wd // 8)], s_in + 1, 0.5)
        with torch.no_grad():
            out = mv.MinVIS.run_window_inference(stub, images, window_size=WIN)
            full = head(backbone(images))
            mine = TR.run_window_inference(images, lambda x: TR.resnet50(x, Wall),
                                           lambda f: TR.frame_decoder(TR.pixel_decoder(f, Wall)[2], TR.pixel_decoder(f, Wall)[0], Wall), WIN)
            out_san = san.SANOnline.run_window_inference(stub, images, mg, window_size=WIN)
            full_san = head(backbone(images), mg)
        ok = all((out[k] - full[k]).abs().max() < 1e-3 for k in ("pred_masks", "pred_embeds")) and \
            (out["pred_masks"] - mine["pred_masks"]).abs().max() < 1e-3 and \
            all((out_san[k] - full_san[k]).abs().max() < 1e-3 for k in ("pred_masks", "pred_embeds", "class_attn_biases"))
        if ok:                                                               # ... and a tracker assignment without near-ties
            with torch.no_grad():
                mine_san = TR.run_window_inference(
                    images, lambda x: TR.resnet50(x, Wall),
                    lambda f, ex: TR.side_frame_decoder(TR.pixel_decoder(f, Wall, extra_features=ex)[2],
                                                        TR.pixel_decoder(f, Wall, extra_features=ex)[0], Wsan, clip_heads=4), WIN, clip_feats=mg)
            ok = torch.equal(TR.video_match_via_embeds(mine["pred_embeds"][0])[0], mv.batch_video_match_via_embeds(out["pred_embeds"])[0][0]) and \
                torch.equal(TR.video_match_via_embeds(mine_san["pred_embeds"][0])[0], mv.batch_video_match_via_embeds(out_san["pred_embeds"])[0][0])
        if ok:
            break
    else:
        raise RuntimeError("no stable seed found")
    print("window fixture uses input seed", s_in)
    with torch.no_grad():
        post = mv.MinVIS.post_processing(stub, dict(out))
        idx, _ = mv.batch_video_match_via_embeds(out["pred_embeds"])
        post_san_idx, _ = mv.batch_video_match_via_embeds(out_san["pred_embeds"])
    np.savez_compressed(os.path.join(GOLD, "window_inference.npz"), spec_bb=_spec_arrays(spec_bb), spec_pd=_spec_arrays(spec_pd),
                        spec_dec=_spec_arrays(spec_dec), spec_sdec=_spec_arrays(spec_sdec),
                        seeds=np.array([171, 172, 173, 174, s_in, s_in + 1]), thw_win=np.array([T, H, Wd, WIN]),
                        pred_embeds=out["pred_embeds"].numpy(), indices=idx.numpy(),   # (un-tracked masks / logits = the tracked ones, un-permuted)
                        post_pred_masks=post["pred_masks"].numpy(), post_pred_logits=post["pred_logits"].numpy(),
                        san_pred_masks=out_san["pred_masks"].numpy(), san_pred_embeds=out_san["pred_embeds"].numpy(),
                        san_class_attn_biases=out_san["class_attn_biases"].numpy(), san_indices=post_san_idx.numpy())
    print("wrote window_inference.npz", out["pred_masks"].shape, out_san["class_attn_biases"].shape)


GENERATORS = {"window": gen_window_inference, "sidevideo": gen_side_video_decoder, "text": gen_clip_text, "swin": gen_swin, "swinape": gen_swin_ape, "msda": gen_msda, "framedec": gen_frame_decoder_and_tracker, "san": gen_side_adapter, "resampler": gen_resampler, "pixdec": gen_pixel_decoder_and_decoder, "clip": gen_clip_visual, "clipmask": gen_clip_visual_mask_prompt,
              "pe": gen_position_encodings, "heads": gen_decoder_head_variants}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
