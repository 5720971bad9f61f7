"""bf16x2 (3-product split: a1 b0 + a0 b1 + a0 b0, 16 significand bits per operand) against bf16x3 (6 products, f32-grade) in the
f32-policy GEMMs: per-query mask IoU vs the f32 CPU oracle at 720p and the time per clip, for the headline model (openvis, where
the pixel decoder + masked-attention decoder run f32) and for SANOnline (backbone + side adapter f32 as well).

  python tools/exp_bf16x2.py > gpurun_out/bf16x2.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from openvis_amd import config, ops, weights
from openvis_amd.catalog import MetadataCatalog
from oracle import torch_ref as TR

K, T = 40, 2
names = [f"class_{i}" for i in range(K)]
MetadataCatalog.get("synthetic_x2").set(thing_classes=names)
text = bench.synth_text(K, 512)
frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
torch.set_num_threads(min(32, torch.get_num_threads()))


def per_query_iou(g, r):
    out = []
    for q in range(g.shape[0]):
        a, b = g[q] > 0, r[q] > 0
        u = (a | b).sum().item()
        out.append(1.0 if u == 0 else (a & b).sum().item() / u)
    return np.array(out)


for arch, spec, fn in (("OpenVIS", weights.openvis_spec("r50", None, 100), TR.openvis_forward),
                       ("SANOnline", weights.san_spec("r50", None, 100), TR.san_online_forward)):
    sd = weights.random_init(spec, seed=42)
    ref_st = {}
    with torch.no_grad():
        fn(frames, sd, text, stages=ref_st)
    r = ref_st["pred_masks"]
    r = r[0] if r.dim() == 5 else r
    for mode in (1, 2):
        ops.set_f32_gemm_mode(mode)
        cfg = config.get_cfg()
        if arch == "SANOnline":
            cfg.MODEL.META_ARCHITECTURE = arch
            cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "SideAdapterFrameMultiScaleMaskedTransformerDecoder"
        model = config.build_model(cfg)
        model.load_state_dict(sd)
        model.clip_adapter.set_text_features(names, text)
        batch = [{"image": [f for f in frames], "dataset_name": "synthetic_x2"}]
        st = {}
        model(batch, stages=st)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            model(batch)
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 100
        g = st["pred_masks"].cpu()
        g = g[0] if g.dim() == 5 else g
        extra = ""
        if "indices" in st:
            same = (st["indices"].cpu().numpy().reshape(ref_st["indices"].shape) == ref_st["indices"].numpy()).all(axis=0)
            extra = f", tracks identical {int(same.sum())}/100"
            scale = float(sd["clip_adapter.clip_model.logit_scale"].exp())
            lg, lr = st["pred_logits"].cpu(), ref_st["pred_logits"]
            extra += f", cosine max err {(lg.reshape(lr.shape) - lr).abs().max().item() / scale:.2e}"
        iou = per_query_iou(g, r)
        print(f"{arch} f32-GEMM mode {mode} ({'bf16x3, 6 products' if mode == 1 else 'bf16x2, 3 products'}): per-query IoU min {iou.min():.5f} "
              f"p10 {np.quantile(iou, 0.1):.5f} median {np.median(iou):.5f}, < 0.999: {(iou < 0.999).sum()}, < 0.9995: {(iou < 0.9995).sum()}"
              f"{extra}; {ms:.1f} ms per 2-frame clip", flush=True)
        del model
        torch.cuda.empty_cache()
ops.set_f32_gemm_mode(1)
