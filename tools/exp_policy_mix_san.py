"""Which stage's fp16 operands cost SANOnline / BriVIS their tracks and mask IoU at 720p (exploration; the counterpart of
tools/exp_policy_mix.py for the online models).  Full architecture (R50 + SideAdapter ViT-B/16), random-init weights seed 42,
T frames of 720x1280, HIP path against the f32 CPU oracle (oracle/torch_ref.py san_online_forward / brivis_forward).

  python tools/exp_policy_mix_san.py [T] > gpurun_out/policy_mix_san.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from openvis_amd import config, weights
from openvis_amd.catalog import MetadataCatalog
from oracle import torch_ref as TR

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = 40
names = [f"class_{i}" for i in range(K)]
MetadataCatalog.get("synthetic_c3").set(thing_classes=names)
text = bench.synth_text(K, 512)
frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
torch.set_num_threads(min(32, torch.get_num_threads()))


def build(arch, mix):
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = arch
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "SideAdapterFrameMultiScaleMaskedTransformerDecoder"
    model = config.build_model(cfg)
    model.backbone.precision = mix["backbone"]
    model.sem_seg_head.predictor.precision = mix["decoder"]
    if mix["clip"] != model.clip_adapter.precision:
        from openvis_amd.modeling.clip_adapter.side_adapter import SideAdapter
        model.clip_adapter = SideAdapter("ViT-B/16", broken_idx=9, merge_ids=[3, 6, 9], num_queries=100, precision=mix["clip"])
    if arch == "BriVIS":
        model.resampler.precision = mix["resampler"]
    return model


for arch, fn in (("SANOnline", TR.san_online_forward), ("BriVIS", TR.brivis_forward)):
    sd = weights.random_init(weights.brivis_spec("r50", None, 100), seed=42)
    ref_st = {}
    t0 = time.time()
    with torch.no_grad():
        ref = fn(frames, sd, text, stages=ref_st)
    print(f"# {arch}: oracle {time.time() - t0:.1f} s for {T} frames")
    r = ref_st["pred_masks"]
    ir = ref_st["indices"].numpy()
    M = lambda b, c, r: {"backbone": b, "decoder": "fp32", "clip": c, "resampler": r}
    mixes = ([M("fp16", "fp16", "fp16"), M("fp16", "fp32", "fp16"), M("fp32", "fp16", "fp16"), M("fp32", "fp32", "fp16")] if arch == "SANOnline" else
             [M("fp16", "fp16", "fp16"), M("fp16", "fp16", "fp32"), M("fp16", "fp32", "fp32"), M("fp32", "fp16", "fp32"),
              M("fp32", "fp16", "fp16"), M("fp32", "fp32", "fp32")])          # first = the round-1 default ("mixed")
    for mix in mixes:
        model = build(arch, mix)
        model.load_state_dict(sd)
        model.clip_adapter.set_text_features(names, text)
        st = {}
        torch.cuda.synchronize()
        t0 = time.time()
        out = model([{"image": [f for f in frames], "dataset_name": "synthetic_c3"}], stages=st)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ig = st["indices"].cpu().numpy().reshape(ir.shape)
        same = (ig == ir).all(axis=0)
        g = st["pred_masks"].cpu()
        if g.dim() == 5:
            g = g[0]
        rr = r[0] if r.dim() == 5 else r
        per_q = []
        for q in range(g.shape[0]):
            if not same[q]:
                continue
            a, b = g[q] > 0, rr[q] > 0
            u = (a | b).sum().item()
            per_q.append(1.0 if u == 0 else (a & b).sum().item() / u)
        per_q = np.array(per_q) if per_q else np.array([0.0])
        sel = torch.from_numpy(np.nonzero(same)[0])
        lg, lr = st["pred_logits"].cpu(), ref_st["pred_logits"]
        lg = lg.reshape(lr.shape) if lg.numel() == lr.numel() else lg
        dl = (lg[..., sel, :] - lr[..., sel, :]).abs().max().item() if len(sel) else float("nan")
        dp = (st["probs"].cpu()[sel] - ref_st["probs"][sel]).abs().max().item() if len(sel) else float("nan")
        scale = float(sd["clip_adapter.clip_model.logit_scale"].exp())
        print(f"{arch} backbone {mix['backbone']} decoder {mix['decoder']} clip {mix['clip']} resampler {mix['resampler']}: "
              f"tracks identical {int(same.sum())}/100, per-query IoU min {per_q.min():.5f} median {np.median(per_q):.5f} "
              f"(< 0.999: {(per_q < 0.999).sum()}), logits max err {dl:.4f} (= {dl / scale:.2e} on the cosine), probs max err {dp:.2e}, "
              f"{dt * 1e3:.0f} ms first call")
        del model
        torch.cuda.empty_cache()
