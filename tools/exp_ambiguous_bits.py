"""Where do the mask bits that differ from the f32 oracle sit?  |oracle logit| at the differing positions, per model policy
(C2: OpenVIS, fp16-operand backbone; C3: SANOnline, f32-class everywhere).  Oracle runs on the host (minutes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from openvis_amd import config, weights
from openvis_amd.catalog import MetadataCatalog
from oracle import torch_ref as TR

torch.set_num_threads(min(32, torch.get_num_threads()))


def report(tag, got, ref):
    d = (got > 0) != (ref > 0)
    a = ref.abs()[d]
    print(f"{tag}: {int(d.sum())} of {d.numel()} bits differ; |oracle logit| there: "
          + (f"max {a.max():.3e} p99 {a.quantile(0.99):.3e} median {a.median():.3e}" if d.any() else "-"),
          "| max |logit diff| overall %.3e" % (got - ref).abs().max(), flush=True)
    for eps in (1e-4, 1e-3, 1e-2, 1e-1):
        print(f"   differing bits with |oracle logit| >= {eps:g}: {int((a >= eps).sum())};  pixels with |oracle logit| < {eps:g}: {int((ref.abs() < eps).sum())}")


def c2(T=2, split="auto"):
    K = 482
    sd = weights.random_init(weights.openvis_spec("r50", None, 100), seed=42)
    cfg = config.get_cfg()
    cfg.MODEL.F32_GEMM_SPLIT = split
    model = config.build_model(cfg)
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_c2").set(thing_classes=names)
    text = bench.synth_text(K, 512, spread=0.25)
    model.clip_adapter.set_text_features(names, text)
    frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
    st, ref_st = {}, {}
    model([{"image": [f for f in frames], "dataset_name": "synthetic_c2"}], stages=st)
    torch.cuda.synchronize()
    images, _ = TR.preprocess([f for f in frames])
    with torch.no_grad():
        feats = TR.resnet50(images, sd)
        mf, _, ms = TR.pixel_decoder(feats, sd)
        _, pm = TR.video_decoder(ms, mf, sd)
    report(f"C2 [{split}] {T} frames", st["pred_masks"][0].cpu(), pm[0])
    # the same model with an f32 backbone: what the split itself contributes
    cfg.MODEL.BACKBONE_PRECISION = "fp32"
    model2 = config.build_model(cfg)
    model2.load_state_dict(sd)
    model2.clip_adapter.set_text_features(names, text)
    st2 = {}
    model2([{"image": [f for f in frames], "dataset_name": "synthetic_c2"}], stages=st2)
    torch.cuda.synchronize()
    report(f"C2 [{split}] f32 backbone", st2["pred_masks"][0].cpu(), pm[0])


def c3(T=2, split="auto"):
    from tests.test_workload_parity_gpu import _build
    model, _ = _build("SANOnline", split=split)
    sd = weights.random_init(weights.san_spec("r50", None, 100), seed=42)
    frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
    st, ref_st = {}, {}
    model([{"image": [f for f in frames], "dataset_name": "synthetic_workload"}], stages=st)
    torch.cuda.synchronize()
    with torch.no_grad():
        TR.san_online_forward(frames, sd, bench.synth_text(40, 512), stages=ref_st)
    report(f"C3 [{split}] {T} frames", st["pred_masks"][0].cpu(), ref_st["pred_masks"][0])


if __name__ == "__main__":
    for what in sys.argv[1:] or ["c3", "c2"]:
        {"c2": c2, "c3": c3}[what]()
