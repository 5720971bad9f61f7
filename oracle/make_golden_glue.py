"""TEST INFRASTRUCTURE ONLY -- fixtures that pin the META-ARCHITECTURE GLUE of the oracle (oracle/torch_ref.py) and of the
HIP path to the REFERENCE's own Python functions, imported from /root/reference in the build container (oracle/ref_import.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_glue.py [functions|forward ...]

Two files:

* tests/golden/glue_functions.npz -- the glue functions called ALONE, as unbound methods on a stub `self`, on crafted inputs:
  OpenVIS.open_vocabulary_inference (openvis.py:110-147) with a stub clip_adapter that returns recorded logits / valid flags,
  VideoMaskFormer.postprocess + inference_video (video_maskformer.py:215-229, 262-298), ClipAdapter._preprocess_image +
  encode_image up to the tower's input (adapter.py:73-116, 140-142), BriVIS.reset_image_output_order + post_processing
  (brivis.py:231-265), MinVIS.post_processing (minvis.py:320-338), batch_index (utils/index.py:4-18).
* tests/golden/glue_forward.npz -- the reference's whole eval `forward` of OpenVIS, OpenVISOnline, SAN, SANOnline and BriVIS
  (openvis.py:47-108, 177-242; san.py:84-144, 177-283; brivis.py:105-211; OpenVIS also with the reference's AdaptedClipAdapter,
  mask_adapted_adapter.py:58-148) called as unbound methods on a stub whose sem_seg_head is the
  reference's own pixel decoder + decoder, whose clip_adapter is the reference's own ClipAdapter / SideAdapter over the vendored
  CLIP at a tiny size, whose resampler is the reference's TemporalInstanceResampler.

What stays third-party and UNPINNED inside these fixtures (absent from /root/reference, restated in oracle/torch_ref.py):
detectron2's ResNet-50 (the stub's backbone), `ImageList.from_tensors`, `BitMasks.get_bounding_boxes`, torchvision's `roi_align`.
`Tensor.cuda()` is identity and `Tensor.half()` is `.float()` while the reference runs (adapter.py:95, 108, 111 hard-wire them; the fixture pins
the glue in f32 on the CPU).  The fixtures are data: inputs (or the seeds that regenerate them) and the reference's outputs."""
import contextlib
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from oracle import ref_import as R
from oracle import torch_ref as TR
from oracle.make_golden import _build_pixel_decoder, _load_synth, _spec_arrays
from oracle import fixtures as FX


# ---------------------------------------------------------------------------------------------------------------------------
# third-party structures the reference's glue touches (restated, unpinned) and the import plumbing
# ---------------------------------------------------------------------------------------------------------------------------
class _ImageList:
    """detectron2.structures.ImageList.from_tensors(tensors, size_divisibility): zero pad right / bottom to a multiple."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        d = size_divisibility
        if d > 1:
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (H, W), pad_value)
        for i, t in enumerate(tensors):
            out[i, ..., :t.shape[-2], :t.shape[-1]] = t
        return _ImageList(out, sizes)


class _BitMasks:
    def __init__(self, tensor):
        self.tensor = tensor

    def get_bounding_boxes(self):
        return types.SimpleNamespace(tensor=TR.bitmask_boxes(self.tensor))


@contextlib.contextmanager
def _cpu_f32():
    """adapter.py:95 `.cuda()` is identity and :108 / :111 `.half()` is `.float()` (the frames arrive as uint8) while the reference's
    glue runs on the CPU in f32."""
    cuda, half = torch.Tensor.cuda, torch.Tensor.half
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.half = lambda self, *a, **k: self.float()
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.Tensor.half = cuda, half


def _ref_modules():
    """Import the reference's meta-architectures through the shims and hand them the restated third-party structures."""
    R.install()
    ca = sys.modules["openvis.modeling.clip_adapter"]
    ad = R.ref("openvis.modeling.clip_adapter.adapter")
    sa = R.ref("openvis.modeling.clip_adapter.side_adapter")
    R.ref("openvis.modeling.clip_adapter.text_prompt")
    ca.SideAdapter = sa.SideAdapter
    ca.ClipAdapter = ad.ClipAdapter
    if not hasattr(ca, "build_clip_adapter"):
        ca.build_clip_adapter = None                     # only from_config calls it
    ov = R.ref("openvis.openvis")
    san = R.ref("openvis.san")
    bv = R.ref("openvis.brivis")
    mv = R.ref("openvis.modeling.minvis")
    vm = R.ref("openvis.modeling.video_maskformer")
    ix = R.ref("openvis.utils.index")
    for m in (ov, san, bv, mv, vm):
        m.ImageList = _ImageList
    ad.BitMasks = _BitMasks
    ad.roi_align = TR.roi_align
    mac = R.ref("mask_adapted_clip.model")
    return types.SimpleNamespace(ad=ad, sa=sa, ov=ov, san=san, bv=bv, mv=mv, vm=vm, ix=ix, mac=mac)


def _pack(bits):
    return np.packbits(np.asarray(bits, dtype=bool).reshape(-1))


# ---------------------------------------------------------------------------------------------------------------------------
# the glue functions alone
# ---------------------------------------------------------------------------------------------------------------------------
def gen_glue_functions():
    M = _ref_modules()
    out = {}
    g = torch.Generator().manual_seed(401)

    # --- OpenVIS.open_vocabulary_inference (openvis.py:110-147): 7 frames = chunks of 5 + 2; recorded crop logits -------------
    T, Q, K, Hp, Wp = 7, 9, 6, 8, 12
    valid = torch.rand(T, Q, generator=g) > 0.45
    valid[:, 2] = False                                  # a query without any crop
    valid[:, 5] = False
    valid[5:, :] = False                                 # the second chunk has no crop at all -> clip_adapter returns None
    valid[3, 7] = True
    crop_logits = torch.randn(int(valid.sum()), K, generator=g) * 3
    masks = torch.randn(Q, T, Hp, Wp, generator=g)
    frames = torch.zeros(T, 3, Hp, Wp)
    calls = []

    def make_adapter(valid, crop_logits):
        def adapter(part_frames, class_names, part_masks):
            t0 = sum(c for c in calls)
            calls.append(len(part_frames))
            v = valid[t0:t0 + len(part_frames)]
            n0 = int(valid[:t0].sum())
            assert part_masks.shape[0] == len(part_frames) and part_masks.shape[1] == Q      # (T, N, H, W) sigmoid probabilities
            if v.sum() == 0:
                return None, v
            return crop_logits[n0:n0 + int(v.sum())], v
        return adapter

    stub = types.SimpleNamespace(device=torch.device("cpu"), clip_adapter=make_adapter(valid, crop_logits))
    names = [f"c{i}" for i in range(K)]
    with torch.no_grad():
        probs, vmasks = M.ov.OpenVIS.open_vocabulary_inference(stub, torch.zeros(Q, 2), masks, frames, names)
    assert calls == [5, 2]
    out.update(ovi_valid=valid.numpy(), ovi_crop_logits=crop_logits.numpy(), ovi_masks=masks.numpy(), ovi_probs=probs.numpy(),
               ovi_masks_out=vmasks.numpy())
    # OpenVISOnline's copy (openvis.py:244-281) walks chunks of 10: same arithmetic, must give the same rows
    calls.clear()
    with torch.no_grad():
        probs_on, vmasks_on = M.ov.OpenVISOnline.open_vocabulary_inference(stub, torch.zeros(Q, 2), masks, frames, names)
    assert calls == [7] and torch.equal(probs_on, probs) and torch.equal(vmasks_on, vmasks)
    # no valid crop in the whole clip -> ([], []) (openvis.py:127-128); no scores at all -> ([], []) (:143-145)
    calls.clear()
    stub0 = types.SimpleNamespace(device=torch.device("cpu"), clip_adapter=make_adapter(torch.zeros_like(valid), crop_logits[:0]))
    e1 = M.ov.OpenVIS.open_vocabulary_inference(stub0, torch.zeros(Q, 2), masks, frames, names)
    e2 = M.ov.OpenVIS.open_vocabulary_inference(stub0, torch.zeros(0, 2), masks, frames, names)
    assert e1 == ([], []) and e2 == ([], [])
    out["ovi_empty_returns_lists"] = np.array([1])

    # --- VideoMaskFormer.postprocess + inference_video (video_maskformer.py:215-229, 262-298) -----------------------------------
    Qv, T, K = 7, 3, 5
    h, w = 16, 24                                        # stride-4 logits of a 64 x 96 padded frame
    img_size, out_hw = (60, 90), (75, 112)
    low = torch.randn(Qv, T, h, w, generator=g) * 2
    cls = torch.softmax(torch.randn(Qv, K, generator=g) * 2, dim=-1)
    head = types.SimpleNamespace(num_classes=K)
    stub = types.SimpleNamespace(sem_seg_head=head)
    with torch.no_grad():
        cls2, up = M.vm.VideoMaskFormer.postprocess(stub, cls, low, (4 * h, 4 * w))
        assert cls2 is cls                               # K columns, not K + 1: no softmax / drop (video_maskformer.py:218-219)
        vo = M.vm.VideoMaskFormer.inference_video(Qv, K, cls2, up, img_size, out_hw[0], out_hw[1])
        vo_same = M.vm.VideoMaskFormer.inference_video(Qv, K, cls2, up, img_size, img_size[0], img_size[1])
        vo_empty = M.vm.VideoMaskFormer.inference_video(Qv, K, [], [], img_size, out_hw[0], out_hw[1])
    assert vo_empty == {"image_size": out_hw, "pred_entropys": [], "pred_scores": [], "pred_labels": [], "pred_masks": []}
    out.update(iv_lowres=low.numpy(), iv_cls=cls.numpy(), iv_sizes=np.array([*img_size, *out_hw]),
               iv_scores=np.array(vo["pred_scores"], np.float32), iv_labels=np.array(vo["pred_labels"], np.int64),
               iv_entropys=np.array(vo["pred_entropys"], np.float32),
               iv_masks=_pack(torch.stack(vo["pred_masks"]).numpy()), iv_masks_shape=np.array(torch.stack(vo["pred_masks"]).shape),
               iv_masks_same=_pack(torch.stack(vo_same["pred_masks"]).numpy()),
               iv_masks_same_shape=np.array(torch.stack(vo_same["pred_masks"]).shape))
    # K + 1 columns: softmax and drop of the last column (the SAN family's form of the same function)
    cls_bg = torch.randn(Qv, K + 1, generator=g)
    with torch.no_grad():
        cls3, _ = M.vm.VideoMaskFormer.postprocess(stub, cls_bg, low, (4 * h, 4 * w))
    out.update(iv_cls_bg=cls_bg.numpy(), iv_cls_bg_out=cls3.numpy())

    # --- ClipAdapter._preprocess_image + encode_image up to the tower (adapter.py:73-116, 140-142) -------------------------------
    T, N, H, W, Hp, Wp, RES = 2, 7, 60, 90, 64, 96, 32
    frames = (torch.rand(T, 3, H, W, generator=g) * 255).floor()            # raw uint8 values, UN-padded (openvis.py:98)
    low = torch.randn(N, T, Hp // 4, Wp // 4, generator=g) * 1.5 - 1.0
    yy, xx = torch.meshgrid(torch.arange(Hp // 4), torch.arange(Wp // 4), indexing="ij")
    for n in range(N):                                                      # blobs: compact objects, some touching the border
        cy, cx = 2 + 2 * n, 3 + 3 * n
        low[n] += 4.0 * torch.exp(-(((yy - cy) / (1.5 + 0.3 * n)) ** 2 + ((xx - cx) / (2.0 + 0.5 * n)) ** 2))
    low[3] = -3.0 - low[3].abs()                                            # an empty mask in every frame
    low[5, 1] = -3.0 - low[5, 1].abs()                                      # ... and one empty in frame 1 only
    low[6, 0, :, :] = -5.0
    low[6, 0, 15, 22:] = 5.0                                                # a wide flat object at the bottom-right corner:
    #                                                                         its square box reaches far beyond the frame
    with torch.no_grad():
        up = F.interpolate(low, size=(Hp, Wp), mode="bilinear", align_corners=False)          # openvis.py:87-96
        part_masks = up.sigmoid().transpose(0, 1).contiguous()                                 # openvis.py:119
    adapter = types.SimpleNamespace(input_resolution=RES)
    captured = {}
    adapter.clip_prep_img = M.ad.Normalize(M.ad.PIXEL_MEAN, M.ad.PIXEL_STD)

    def visual(image):
        captured["tower_input"] = image
        return torch.ones(image.shape[0], 4)
    adapter.clip_model = types.SimpleNamespace(visual=visual)
    adapter.normalize = lambda feat: feat
    with torch.no_grad(), _cpu_f32():
        regions, pvalid = M.ad.ClipAdapter._preprocess_image(adapter, frames, part_masks)
        M.ad.ClipAdapter.encode_image(adapter, regions)
        none_regions, none_valid = M.ad.ClipAdapter._preprocess_image(adapter, frames, torch.zeros_like(part_masks))
    assert none_regions is None and not bool(none_valid.any())
    out.update(pp_frames=frames.numpy().astype(np.uint8), pp_lowres=low.numpy(), pp_sizes=np.array([H, W, Hp, Wp, RES]),
               pp_valid=pvalid.numpy(), pp_regions=regions.numpy(), pp_tower_input=captured["tower_input"].numpy())

    # --- batch_index, MinVIS.post_processing, BriVIS.reset_image_output_order + post_processing ---------------------------------
    T, Q, C, h, w, K = 4, 6, 8, 6, 8, 3                # (stride-4 maps of the model have h w % 4 == 0)
    embeds = torch.randn(1, T, Q, C, generator=g)
    logits = torch.randn(1, T, Q, K + 1, generator=g)
    pmasks = torch.randn(1, Q, T, h, w, generator=g)
    src = torch.randn(3, 5, 4, generator=g)
    idx = torch.stack([torch.randperm(5, generator=g)[:4] for _ in range(3)])
    out.update(bi_src=src.numpy(), bi_idx=idx.numpy(), bi_first=M.ix.batch_index(src, idx).numpy(),
               bi_second=M.ix.batch_index(src.transpose(0, 1).contiguous(), idx.t().contiguous(), batch_first=False).numpy())
    with torch.no_grad():
        post = M.mv.MinVIS.post_processing(types.SimpleNamespace(), dict(pred_logits=logits, pred_masks=pmasks, pred_embeds=embeds))
        indices, frame_embeds = M.mv.batch_video_match_via_embeds(embeds)
        bstub = types.SimpleNamespace(sem_seg_head=types.SimpleNamespace(num_classes=K))
        re = M.bv.BriVIS.reset_image_output_order(bstub, dict(pred_logits=logits.clone(), pred_masks=pmasks.clone()), indices)
        cls_b, mask_b = M.bv.BriVIS.post_processing(bstub, dict(pred_logits=logits.clone(), pred_masks=pmasks.clone()), (4 * h, 4 * w))
    out.update(tr_embeds=embeds.numpy(), tr_logits=logits.numpy(), tr_masks=pmasks.numpy(), tr_indices=indices.numpy(),
               tr_frame_embeds=frame_embeds.numpy(), tr_post_logits=post["pred_logits"].numpy(), tr_post_masks=post["pred_masks"].numpy(),
               tr_reset_logits=re["pred_logits"].numpy(), tr_reset_masks=re["pred_masks"].numpy(),
               bv_cls=cls_b.numpy(), bv_masks=mask_b.numpy())
    np.savez_compressed(os.path.join(GOLD, "glue_functions.npz"), **out)
    print("wrote glue_functions.npz", {k: v.shape for k, v in out.items()})


# ---------------------------------------------------------------------------------------------------------------------------
# the reference's whole eval forward on stubs
# ---------------------------------------------------------------------------------------------------------------------------
from tests._synth import GLUE_T, GLUE_H, GLUE_W, GLUE_K, GLUE_Q, GLUE_OUT_HW, GLUE_CLIP, glue_frames, glue_text   # noqa: E402


class _Backbone(nn.Module):
    """detectron2's ResNet-50 is not in /root/reference: the stub's backbone is the oracle's restatement (unpinned)."""

    def __init__(self, W):
        super().__init__()
        self.W = W

    def forward(self, x):
        with torch.no_grad():
            return TR.resnet50(x, self.W)


class _Head(nn.Module):
    """MaskFormerHead.forward (mask_former_head.py:113-135) for TRANSFORMER_IN_FEATURE = multi_scale_pixel_decoder: the reference's
    own pixel decoder and decoder."""

    def __init__(self, pd, dec):
        super().__init__()
        self.pixel_decoder, self.predictor, self.num_classes = pd, dec, 1

    def forward(self, features, extra_feats=None):
        with torch.no_grad():
            if extra_feats is not None:
                mf, _, ms = self.pixel_decoder.forward_features(features, extra_feats)
            else:
                mf, _, ms = self.pixel_decoder.forward_features(features)
            return self.predictor(ms, mf)


def _stub(M, Wbb, head, adapter, names):
    s = types.SimpleNamespace(training=False, device=torch.device("cpu"), size_divisibility=32, num_queries=GLUE_Q,
                              pixel_mean=torch.tensor(TR.PIXEL_MEAN).view(-1, 1, 1), pixel_std=torch.tensor(TR.PIXEL_STD).view(-1, 1, 1),
                              backbone=_Backbone(Wbb), sem_seg_head=head, clip_adapter=adapter, window_inference=False, num_frames=GLUE_T,
                              get_class_name_list=lambda dataset_name: names)
    return s


def gen_glue_forward():
    from tests._synth import synth_weights
    from openvis_amd import weights as PW                                  # key / shape list of the R50 backbone only
    M = _ref_modules()
    fd = R.ref("openvis.modeling.transformer_decoder.frame_mask2former_transformer_decoder")
    vd = R.ref("openvis.modeling.transformer_decoder.video_mask2former_transformer_decoder")
    sfd = R.ref("openvis.modeling.transformer_decoder.side_adapter_frame_mask2former_transformer_decoder")
    svd = R.ref("openvis.modeling.transformer_decoder.side_adapter_video_mask2former_transformer_decoder")
    maa = R.ref("openvis.modeling.clip_adapter.mask_adapted_adapter")
    maa.BitMasks, maa.roi_align = _BitMasks, TR.roi_align
    rs = R.ref("openvis.modeling.resampler")
    Q, T, K = GLUE_Q, GLUE_T, GLUE_K
    names = [f"class_{i}" for i in range(K)]
    tiny = dict(width=64, layers=1, heads=1, patch=16, resolution=32, embed_dim=16)
    spec_bb = [(k[len("backbone."):], s) for k, s in PW.openvis_spec("r50", tiny, Q) if k.startswith("backbone.")]
    Wbb = synth_weights(spec_bb, 411, "backbone.")
    dkw = dict(in_channels=256, mask_classification=True, num_classes=1, hidden_dim=256, num_queries=Q, nheads=8, dim_feedforward=2048,
               dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T)
    pd = _build_pixel_decoder()
    spec_pd = _load_synth(pd, 412)
    vdec = vd.VideoMultiScaleMaskedTransformerDecoder(**dkw).eval()
    spec_vdec = _load_synth(vdec, 413)
    fdec = fd.FrameMultiScaleMaskedTransformerDecoder(**dkw).eval()
    spec_fdec = _load_synth(fdec, 414)
    sdec = sfd.SideAdapterFrameMultiScaleMaskedTransformerDecoder(clip_heads=4, **dkw).eval()
    spec_sdec = _load_synth(sdec, 415)
    svdec = svd.SideAdapterVideoMultiScaleMaskedTransformerDecoder(clip_heads=4, **dkw).eval()
    spec_svdec = _load_synth(svdec, 4151)
    M.ad.build_clip_model = lambda name: M.mac.CLIP(**GLUE_CLIP)            # clip.load() needs the network
    M.sa.build_clip_model = lambda name: M.mac.CLIP(**GLUE_CLIP)
    cad = M.ad.ClipAdapter("tiny", text_templates=["{}"]).eval()
    spec_cad = _load_synth(cad, 416)
    maa.build_mask_adapted_clip_model = lambda name, depth: M.mac.CLIP(**{**GLUE_CLIP, "mask_prompt_depth": depth})
    acad = maa.AdaptedClipAdapter("tiny", 3, True, text_templates=["{}"]).eval()           # mask_prompt_depth 3, mask_prompt_fwd True
    spec_acad = _load_synth(acad, 4161)
    sad = M.sa.SideAdapter("tiny", out_dims=256, broken_idx=3, merge_ids=[1, 2, 3], num_queries=Q, text_templates=["{}"]).eval()
    spec_sad = _load_synth(sad, 417)
    LOGIT_SCALE = float(np.log(1 / 0.07))                                   # CLIP's own initial value (model.py: logit_scale); the synthetic
    sad.clip_model.logit_scale.data.fill_(LOGIT_SCALE)                      # N(0,1) draw would leave every softmax uniform to 1e-7
    res = rs.TemporalInstanceResampler(hidden_dim=256, feed_dim=2048, nheads=8, nlayers=6).eval()
    spec_res = _load_synth(res, 418)
    # Round 6: a label space on which the classification can FAIL (oracle/fixtures.py).  (i) the ClipAdapter / AdaptedClipAdapter towers get
    # peaked attention (weights.sharpen_clip_attention: q and k rows of every in_proj x CLIP_QK_GAIN, applied to the reference modules HERE
    # and to the oracle's / product's weight dicts wherever they are rebuilt from the specs -- tests/test_oracle_glue.py load_glue_forward), so
    # that crop embeddings differ between queries; the SideAdapter stays as it is (its CLIP features feed the pixel decoder: masks, tracker and
    # their stable seeds would move).  (ii) per architecture the reference's forward runs TWICE: once on glue_text to capture the image features
    # it hands to cal_sim_logits, then on text rows built from those features (ten winners, >= 5 distinct labels, graded scores).
    CLIP_QK_GAIN = 4.0
    SHARP_ARCHS = ("openvis", "openvis_adapted")           # OpenVIS (offline decoder) with ClipAdapter and with AdaptedClipAdapter: the headline meta-architecture
    with torch.no_grad():
        for ad_ in (cad, acad):
            for n_, p_ in ad_.named_parameters():
                if n_.startswith("clip_model.visual.") and (n_.endswith("attn.in_proj_weight") or n_.endswith("attn.in_proj_bias")):
                    p_[: 2 * (p_.shape[0] // 3)] *= CLIP_QK_GAIN
    # (iii) the decoders' mask logits get a usable range: the synthetic mask_embed MLP leaves |logit| ~ 1e-2, i.e. sigmoid = 0.5 +- 0.003 for every
    # query, every soft mask -- and with it every masked crop -- the same picture.  The last mask_embed layer x MASK_GAIN (reference modules here,
    # weight dicts in W_of / load_glue_forward) spreads the masks (and makes every thresholded bit LESS ambiguous, not more).
    MASK_GAIN = 100.0
    with torch.no_grad():
        for dec_ in (vdec,):                              # the offline OpenVIS decoder only: the per-frame / side-adapter decoders feed trackers and
            dec_.mask_embed.layers[2].weight *= MASK_GAIN   # fp16-stored tracked-mask fixtures whose tolerances are absolute
            dec_.mask_embed.layers[2].bias *= MASK_GAIN
    plain_text = glue_text(419, GLUE_CLIP["embed_dim"])
    feats_seen = []

    def set_text(text):
        cad.text_cache = dict(zip(names, text))                             # encode_text (adapter.py:121-138): every word cached
        acad.text_cache = dict(zip(names, text))
        sad.text_cache = dict(zip([n.replace("_", " ") for n in names], text))   # side_adapter.py:214 strips "()_" before the lookup

    for ad_ in (cad, acad, sad):
        def wrap(orig):
            def f(text_feats, image_feats, *a, **k):
                feats_seen.append((text_feats.detach().clone(), image_feats.detach().clone()))
                return orig(text_feats, image_feats, *a, **k)
            return f
        ad_.cal_sim_logits = wrap(ad_.cal_sim_logits)
    set_text(plain_text)
    out = dict(spec_bb=_spec_arrays(spec_bb), spec_pd=_spec_arrays(spec_pd), spec_vdec=_spec_arrays(spec_vdec),
               spec_fdec=_spec_arrays(spec_fdec), spec_sdec=_spec_arrays(spec_sdec), spec_cad=_spec_arrays(spec_cad),
               spec_sad=_spec_arrays(spec_sad), spec_res=_spec_arrays(spec_res), spec_svdec=_spec_arrays(spec_svdec), spec_acad=_spec_arrays(spec_acad),
               seeds=np.array([411, 412, 413, 414, 415, 416, 417, 418, 419]), seeds_more=np.array([4151, 4161]),
               dims=np.array([T, GLUE_H, GLUE_W, K, Q, *GLUE_OUT_HW]), side_logit_scale=np.array([LOGIT_SCALE], np.float32),
               clip_qk_gain=np.array([CLIP_QK_GAIN], np.float32), mask_gain=np.array([MASK_GAIN], np.float32))

    def W_of(dec_spec, dec_seed, ad_spec, ad_seed, with_res=False):
        W = dict(Wbb)
        W.update(synth_weights(spec_pd, 412, "sem_seg_head.pixel_decoder."))
        W.update(synth_weights(dec_spec, dec_seed, "sem_seg_head.predictor."))
        W.update(synth_weights(ad_spec, ad_seed, "clip_adapter."))
        if ad_spec is not spec_sad:
            W = PW.sharpen_clip_attention(W, CLIP_QK_GAIN)                # (OpenVISOnline's ClipAdapter tower too: same module object)
        if dec_spec is spec_vdec:
            for k_ in ("weight", "bias"):
                W[f"sem_seg_head.predictor.mask_embed.layers.2.{k_}"] = W[f"sem_seg_head.predictor.mask_embed.layers.2.{k_}"] * MASK_GAIN
        if with_res:
            W.update(synth_weights(spec_res, 418, "resampler."))
        if "clip_adapter.bg_embed" in W:
            W["clip_adapter.clip_model.logit_scale"] = torch.tensor(LOGIT_SCALE)
        return W

    def record(stub, key, rec):
        fn = getattr(stub, key)

        def wrapped(*a, **k):
            r = fn(*a, **k)
            rec.setdefault(key, []).append(r)
            return r
        setattr(stub, key, wrapped)

    rois_seen = []

    def roi_rec(inp, rois, output_size, **kw):                        # every roi_align call of the reference: rois [M, 5] = (index, x0, y0, x1, y1)
        rois_seen.append((tuple(inp.shape), rois.clone()))
        return TR.roi_align(inp, rois, output_size, **kw)
    M.ad.roi_align = roi_rec
    maa.roi_align = roi_rec

    def run(arch, seed, text):
        rois_seen.clear()
        feats_seen.clear()
        set_text(text)
        frames = glue_frames(seed)
        inp = [{"image": [f for f in frames], "dataset_name": "glue_val", "height": GLUE_OUT_HW[0], "width": GLUE_OUT_HW[1]}]
        rec = {}
        if arch == "san":                                                     # SAN.forward (san.py:84-144): offline side-adapter decoder, no tracker
            stub = _stub(M, Wbb, _Head(pd, svdec), sad, names)
            stub.postprocess = types.MethodType(M.vm.VideoMaskFormer.postprocess, stub)
            stub.inference_video = M.vm.VideoMaskFormer.inference_video
            W = W_of(spec_svdec, 4151, spec_sad, 417)
            iv = stub.inference_video

            def iv_rec(*a):
                rec["iv_probs"] = [x for x in a if torch.is_tensor(x) and x.dim() == 2][0]
                return iv(*a)
            stub.inference_video = iv_rec
            with torch.no_grad(), _cpu_f32():
                vo = M.san.SAN.forward(stub, inp)
                st = {}
                mine = TR.san_forward(frames, W, text, out_hw=GLUE_OUT_HW, stages=st, broken_idx=3, merge_ids=(1, 2, 3),
                                      resolution=GLUE_CLIP["image_resolution"], clip_heads=4, num_queries=Q)
            return vo, rec, mine, st
        if arch in ("openvis", "openvis_online", "openvis_adapted"):
            cls = M.ov.OpenVISOnline if arch == "openvis_online" else M.ov.OpenVIS
            stub = _stub(M, Wbb, _Head(pd, fdec if arch == "openvis_online" else vdec), acad if arch == "openvis_adapted" else cad, names)
            stub.open_vocabulary_inference = types.MethodType(cls.open_vocabulary_inference, stub)
            if arch != "openvis_online":
                stub.inference_video = M.vm.VideoMaskFormer.inference_video
            else:
                stub.inference_video = types.MethodType(M.mv.MinVIS.inference_video, stub)
                stub.post_processing = types.MethodType(M.mv.MinVIS.post_processing, stub)
                record(stub, "post_processing", rec)
            real = stub.clip_adapter
            stub.clip_adapter = lambda *a: rec.setdefault("clip", []).append(real(*a)) or rec["clip"][-1]
            record(stub, "open_vocabulary_inference", rec)
            W = W_of(spec_fdec if arch == "openvis_online" else spec_vdec, 414 if arch == "openvis_online" else 413,
                     spec_acad if arch == "openvis_adapted" else spec_cad, 4161 if arch == "openvis_adapted" else 416)
            oracle_fn = TR.openvis_online_forward if arch == "openvis_online" else TR.openvis_forward
            okw = dict(clip_heads=GLUE_CLIP["vision_width"] // 64, clip_resolution=GLUE_CLIP["image_resolution"])
            if arch == "openvis_adapted":
                okw.update(mask_prompt_depth=3, mask_prompt_fwd=True)
        else:
            cls = M.san.SANOnline if arch == "san_online" else M.bv.BriVIS
            stub = _stub(M, Wbb, _Head(pd, sdec), sad, names)
            stub.inference_video = types.MethodType(M.mv.MinVIS.inference_video, stub)
            if arch == "san_online":
                stub.post_processing = types.MethodType(M.mv.MinVIS.post_processing, stub)
            else:
                stub.post_processing = types.MethodType(M.bv.BriVIS.post_processing, stub)
                stub.reset_image_output_order = types.MethodType(M.bv.BriVIS.reset_image_output_order, stub)
                stub.resampler = res
            record(stub, "post_processing", rec)
            W = W_of(spec_sdec, 415, spec_sad, 417, with_res=arch == "brivis")
            oracle_fn = TR.san_online_forward if arch == "san_online" else TR.brivis_forward
            okw = dict(broken_idx=3, merge_ids=(1, 2, 3), resolution=GLUE_CLIP["image_resolution"], clip_heads=4, num_queries=Q)
        iv = stub.inference_video

        def iv_rec(*a):
            rec["iv_probs"] = [x for x in a if torch.is_tensor(x) and x.dim() == 2][0]
            return iv(*a)
        stub.inference_video = iv_rec
        with torch.no_grad(), _cpu_f32():
            vo = cls.forward(stub, inp)
            st = {}
            mine = oracle_fn(frames, W, text, out_hw=GLUE_OUT_HW, stages=st, **okw)
        return vo, rec, mine, st

    def rows_of(vo, probs):
        """inference_video does not return the query rows of its top-10: recover them from the probabilities it was given (exact
        float match of score and label; None when two rows hold the same value)."""
        rows = []
        for s_, l_ in zip(vo["pred_scores"], vo["pred_labels"]):
            r = torch.nonzero(probs[:, l_] == torch.tensor(s_, dtype=probs.dtype))[:, 0]
            if len(r) != 1:
                return None
            rows.append(int(r[0]))
        return rows

    def stable(arch, vo, rec, mine, st):
        """the oracle (different summation orders) lands on the reference's side of every threshold: identical top-10 (query row,
        label) sets with scores within 1e-4, identical output masks, identical valid flags."""
        dbg = os.environ.get("GLUE_DEBUG")
        if len(vo["pred_scores"]) != 10 or len(mine["pred_scores"]) != 10:
            if dbg: print("   stable: not 10 outputs")
            return False
        probs = rec["iv_probs"]
        rows = rows_of(vo, probs)
        if rows is None or probs.shape != st["probs"].shape or (probs - st["probs"]).abs().max() > 1e-4:
            if dbg: print("   stable: rows", rows is None, "shapes", probs.shape, st["probs"].shape, "probs diff", float((probs - st["probs"]).abs().max()) if probs.shape == st["probs"].shape else None)
            return False
        a = {(r, l): (s, m) for r, l, s, m in zip(rows, vo["pred_labels"], vo["pred_scores"], vo["pred_masks"])}
        b = {(r, l): (s, m) for r, l, s, m in zip(mine["rows"], mine["pred_labels"], mine["pred_scores"], mine["pred_masks"])}
        if len(a) != 10 or set(a) != set(b) or any(not torch.equal(a[k][1], b[k][1]) or abs(a[k][0] - b[k][0]) > 1e-4 for k in a):
            if dbg: print("   stable: top-10 sets", len(a), set(a) == set(b), [int((a[k][1] != b[k][1]).sum()) for k in a if k in b])
            return False
        if arch in ("openvis", "openvis_online", "openvis_adapted"):
            v = torch.cat([c[1] for c in rec["clip"]])
            if not torch.equal(v, st["valid"]):
                return False
        if isinstance(rec.get("post_processing", [None])[0], dict):          # MinVIS.post_processing: the tracked low-res masks agree, so the
            post = rec["post_processing"][0]                                  # oracle's assignment (st["indices"], stored) is the reference's
            if (post["pred_masks"] - st["pred_masks"]).abs().max() > 1e-3 or (post["pred_logits"] - st.get("pred_logits", post["pred_logits"])).abs().max() > 1e-3:
                if dbg: print("   stable: tracked masks diff", float((post["pred_masks"] - st["pred_masks"]).abs().max()), "logits", float((post["pred_logits"] - st.get("pred_logits", post["pred_logits"])).abs().max()))
                return False
        rec["rows"] = rows
        return True

    for arch, seed0 in (("openvis", 421), ("openvis_online", 521), ("san_online", 621), ("brivis", 721), ("san", 821), ("openvis_adapted", 921)):
        sharp = arch in SHARP_ARCHS
        for seed in range(seed0, seed0 + 200, 10):
            # pass 1 on glue_text: the image features the reference hands to cal_sim_logits -> per-query mean embeddings -> the label space
            vo, rec, mine, st = run(arch, seed, plain_text)
            text2, margin = plain_text, float("nan")
            if not sharp:
                if stable(arch, vo, rec, mine, st):
                    break
                print(arch, "seed", seed, "not stable")
                continue
            if not feats_seen:
                print(arch, "seed", seed, "no crop")
                continue
            valid = torch.cat([c[1] for c in rec["clip"]])
            _, Mq = FX.per_query_mean(torch.cat([f[1] for f in feats_seen]), valid)
            try:
                parts, rep = FX.sharp_parts(Mq, K, seed=seed, n_labels=5, scale=100.0)
            except RuntimeError as e:
                print(arch, "seed", seed, e)
                continue
            text2 = FX.text_from_parts(parts, K)
            vo, rec, mine, st = run(arch, seed, text2)                        # pass 2: the reference's forward on that label space
            if not stable(arch, vo, rec, mine, st):
                print(arch, "seed", seed, "not stable")
                continue
            fl = rec["iv_probs"].flatten().sort(descending=True).values
            margin, distinct = float(fl[9] - fl[10]), len(set(vo["pred_labels"]))
            if margin >= 1e-2 and distinct >= 5 and float(fl[0]) < 0.985:
                break
            print(arch, "seed", seed, f"label space too flat on the reference's own probabilities: margin {margin:.3g}, {distinct} labels, top {float(fl[0]):.3f}")
        else:
            raise RuntimeError(f"no stable seed found for {arch}")
        print(arch, "uses frame seed", seed, "scores", np.round(vo["pred_scores"], 4), "labels", vo["pred_labels"], "margin 10th - 11th %.3f" % margin)
        out[arch + "_text"] = text2.numpy().astype(np.float32)
        out[arch + "_margin"] = np.array([margin])
        p = arch + "_"
        masks = torch.stack(vo["pred_masks"])
        assert tuple(masks.shape) == (10, T) + GLUE_OUT_HW and vo["image_size"] == GLUE_OUT_HW
        out.update({p + "frame_seed": np.array([seed]), p + "rows": np.array(rec["rows"], np.int64), p + "probs": rec["iv_probs"].numpy(),
                    p + "scores": np.array(vo["pred_scores"], np.float32),
                    p + "labels": np.array(vo["pred_labels"], np.int64), p + "entropys": np.array(vo["pred_entropys"], np.float32),
                    p + "masks": _pack(masks.numpy())})
        if arch in ("openvis", "openvis_online", "openvis_adapted"):
            # the square crop boxes the reference handed to roi_align for the FRAME crops (adapter.py:97-108): [M, 4] in (frame, query) order
            out[p + "boxes"] = torch.cat([r[:, 1:] for shp, r in rois_seen if shp[1] == 3]).numpy().astype(np.float32)
            probs, vmasks = rec["open_vocabulary_inference"][0]
            out.update({p + "valid": torch.cat([c[1] for c in rec["clip"]]).numpy(),
                        p + "crop_logits": torch.cat([c[0] for c in rec["clip"] if c[0] is not None]).numpy()})
            assert torch.equal(probs, rec["iv_probs"])
        if "post_processing" in rec:
            post = rec["post_processing"][0]
            if isinstance(post, dict):                                       # MinVIS.post_processing: tracked logits / masks (low-res), the per-frame query embeddings it matched on
                out.update({p + "tracked_logits": post["pred_logits"].numpy(), p + "tracked_masks": post["pred_masks"].numpy().astype(np.float16),
                            p + "pred_embeds": post["pred_embeds"].numpy()})
            else:                                                            # BriVIS.post_processing: (class probabilities, upsampled masks)
                out.update({p + "cls": post[0].numpy()})
        if arch in ("openvis_online", "san_online", "brivis"):
            out[p + "indices"] = st["indices"].numpy()                      # the tracker's assignment (stable(): the outputs built on it agree)
    np.savez_compressed(os.path.join(GOLD, "glue_forward.npz"), **out)
    print("wrote glue_forward.npz", {k: v.shape for k, v in out.items() if not k.startswith("spec")})


GENERATORS = {"functions": gen_glue_functions, "forward": gen_glue_forward}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    for n in sys.argv[1:] or list(GENERATORS):
        GENERATORS[n]()
