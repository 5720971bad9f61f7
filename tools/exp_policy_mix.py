"""Which stage's fp16 operands cost the mask IoU at 720p: backbone vs decoder (exploration; prints per-output-mask IoU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from openvis_amd import config, weights
from openvis_amd.catalog import MetadataCatalog
from oracle import torch_ref as TR

K, T = 40, 2
sd = weights.random_init(weights.openvis_spec("r50", None, 100), seed=42)
names = [f"class_{i}" for i in range(K)]
MetadataCatalog.get("synthetic_c2").set(thing_classes=names)
text = bench.synth_text(K, 512)
frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
ref_st = {}
with torch.no_grad():
    ref = TR.openvis_forward(frames, sd, text, stages=ref_st)
r = ref_st["pred_masks"]
for bb, dec in (("fp16", "fp16"), ("fp32", "fp16"), ("fp16", "fp32"), ("fp32", "fp32")):
    cfg = config.get_cfg()
    model = config.build_model(cfg)
    model.backbone.precision = bb
    model.sem_seg_head.predictor.precision = dec
    model.load_state_dict(sd)
    model.clip_adapter.set_text_features(names, text)
    st = {}
    out = model([{"image": [f for f in frames], "dataset_name": "synthetic_c2"}], stages=st)
    g = st["pred_masks"].cpu()
    iou_all = ((g > 0) & (r > 0)).sum().item() / max(((g > 0) | (r > 0)).sum().item(), 1)
    per_q = []
    for q in range(100):
        a, b = g[0, q] > 0, r[0, q] > 0
        u = (a | b).sum().item()
        per_q.append(1.0 if u == 0 else (a & b).sum().item() / u)
    per_q = np.array(per_q)
    print(f"backbone {bb} decoder {dec}: all-query IoU {iou_all:.5f}; per-query min {per_q.min():.5f} p10 {np.quantile(per_q, 0.1):.5f} "
          f"median {np.median(per_q):.5f}; queries < 0.999: {(per_q < 0.999).sum()}")
