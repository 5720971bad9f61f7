"""fp16 storage of the backbone activations, emulated (OVIS_EXP_BB16=1): per-query mask IoU at 720p against the f32 oracle.
Result: profiles/r02/negative_backbone_fp16_activations.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from openvis_amd import config, weights
from openvis_amd.catalog import MetadataCatalog
from oracle import torch_ref as TR
if os.environ.get("OVIS_EXP_BB16"):          # emulate fp16 activation storage: round every conv output of the ResNet to fp16
    from openvis_amd.modeling.backbone import resnet as _rn
    _cls = next(v for v in vars(_rn).values() if isinstance(v, type) and hasattr(v, "_conv"))
    _orig = _cls._conv
    _cls._conv = lambda self, *a, **k: _orig(self, *a, **k).half().float()
K, T = 40, 2
sd = weights.random_init(weights.openvis_spec("r50", None, 100), seed=42)
names = [f"class_{i}" for i in range(K)]
MetadataCatalog.get("synthetic_c2").set(thing_classes=names)
text = bench.synth_text(K, 512)
for seed in (3, 11):
    frames = bench.synth_frames(T, 720, 1280, seed, "cpu")
    ref_st = {}
    with torch.no_grad():
        ref = TR.openvis_forward(frames, sd, text, stages=ref_st)
    r = ref_st["pred_masks"]
    cfg = config.get_cfg()
    model = config.build_model(cfg)
    model.load_state_dict(sd)
    model.clip_adapter.set_text_features(names, text)
    st = {}
    out = model([{"image": [f for f in frames], "dataset_name": "synthetic_c2"}], stages=st)
    g = st["pred_masks"].cpu()
    per_q = []
    for q in range(100):
        a, b = g[0, q] > 0, r[0, q] > 0
        u = (a | b).sum().item()
        per_q.append(1.0 if u == 0 else (a & b).sum().item() / u)
    per_q = np.array(per_q)
    rows_ref = ref_st["valid"].any(0).nonzero()[:, 0].tolist()
    sg = {(q, l): i for i, (q, l) in enumerate(zip(out["pred_queries"], out["pred_labels"]))}
    sr = {(rows_ref[q], l): i for i, (q, l) in enumerate(zip(ref["rows"], ref["pred_labels"]))}
    ious = []
    for k in set(sg) & set(sr):
        a, b = out["pred_masks"][sg[k]].cpu().numpy().astype(bool), np.asarray(ref["pred_masks"][sr[k]]).astype(bool)
        u = (a | b).sum(); ious.append(1.0 if u == 0 else (a & b).sum() / u)
    print(f"BB16={os.environ.get('OVIS_EXP_BB16')} seed {seed}: per-query IoU min {per_q.min():.5f} p10 {np.quantile(per_q,0.1):.5f} median {np.median(per_q):.5f} <0.999: {(per_q<0.999).sum()}; output masks min {min(ious):.5f} ({len(ious)} common)", flush=True)
