"""Worker of tests/test_sharded_gpu.py: one rank of ONE OpenVIS clip split over several ranks (OpenVIS row: per-frame stages on
the rank's own frames, the offline decoder's cross-attention as split-KV with one all-gather of flash partials per layer).  gloo rendezvous
(OVIS_SPLIT_BACKEND=nccl with one rank: RCCL), all ranks on cuda:0 -- the GPU box has one device.

OVIS_SPLIT_CASE = "c2s":   BASELINE configs[1] at full size (5 frames, 720p, 482 classes) on the separated label space of
                           tests/golden/c2_sharp_classes.npz -- the test holds the result against the ORACLE's golden values;
                  "small": 5 frames at 192x256, 40 classes -- the test holds N ranks against one."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    case = os.environ.get("OVIS_SPLIT_CASE", "small")
    import bench
    from openvis_amd import config, weights, distributed as D
    from openvis_amd.catalog import MetadataCatalog
    torch.cuda.set_device(0)
    rank, world, _ = D.init_from_env(os.environ.get("OVIS_SPLIT_BACKEND", "gloo"), force=True)
    sd = weights.sharpen_clip_attention(weights.random_init(weights.openvis_spec("r50", None, 100), seed=42))
    cfg = config.get_cfg()
    cfg.MODEL.CLIP_ADAPTER.PRECISION = "fp32"
    if case == "c2s":
        from oracle import fixtures as FX                    # the label space is DATA of the golden file; nothing of the oracle computes here
        g = np.load(os.path.join(ROOT, "tests", "golden", "c2_sharp_classes.npz"))
        K, T, H, W, seed = 482, 5, 720, 1280, 1000
        cfg.MODEL.BACKBONE_PRECISION = "fp32"                # as tests/test_c2_720p_gpu.py::test_c2_full_size_classification_on_a_separated_label_space
        text = FX.text_from_parts(FX.parts_from_arrays(g), K)
    else:
        K, T, H, W, seed = 40, 5, 192, 256, 3
        cfg.MODEL.PRECISION = "fp32"
        text = bench.synth_text(K, 512, spread=0.25)
    cfg.MODEL.CLIP_ADAPTER.CROP_LIST = os.environ.get("OVIS_SPLIT_CROP_LIST", "auto")
    model = config.build_model(cfg)
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_split").set(thing_classes=names)
    model.clip_adapter.set_text_features(names, text)
    frames = bench.synth_frames(T, H, W, seed, "cpu")
    inp = [{"image": [f for f in frames], "dataset_name": "synthetic_split"}]
    fr = D.inference_shard(T, rank, world)
    split = os.environ.get("OVIS_SPLIT_OFF") != "1"
    gather_to = int(os.environ["OVIS_GATHER_TO"]) if "OVIS_GATHER_TO" in os.environ else None
    st = {}
    D.SPANS = {}
    if split:
        out = model(inp, stages=st, frame_range=(fr.start, fr.stop), gather_masks_to=gather_to)
    else:
        out = model(inp, stages=st)
    torch.cuda.synchronize()
    pm = st["pred_masks"][0].cpu()                                         # [Q, t_local, h, w] logits
    res = {"rank": rank, "world": world, "range": [fr.start, fr.stop] if split else [0, T], "labels": list(out["pred_labels"]),
           "scores": list(out["pred_scores"]), "queries": list(out["pred_queries"]), "entropys": list(out["pred_entropys"]),
           "probs": st["probs"].clamp(min=0).cpu().tolist(), "pred_logits": st["pred_logits"].cpu().flatten().tolist(),
           "mask_sums": [int(m.sum()) for m in out["pred_masks"]],
           "mask_shape": list(out["pred_masks"][0].shape) if len(out["pred_masks"]) else [],
           "frame_sums": [int(v) for v in torch.stack(list(out["pred_masks"])).sum(dim=(0, 2, 3))] if len(out["pred_masks"]) else [],
           "mask_frames": list(out.get("pred_masks_frames", [])), "spans": D.spans_ms(), "backend": D.backend_name()}
    np.save(f"{out_path}.{rank}.masks.npy", pm.numpy())
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    D.barrier()


if __name__ == "__main__":
    main()
