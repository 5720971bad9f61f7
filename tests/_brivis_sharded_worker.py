"""Worker of tests/test_sharded_gpu.py: one rank of a frame-sharded BriVIS (OVIS_SHARD_ARCH: SANOnline, OpenVISOnline) run (gloo rendezvous, all ranks on cuda:0 --
the GPU box has one device; the exchange logic is what is under test, RCCL itself is the driver's multi-GPU bench)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    import bench
    from openvis_amd import config, weights, distributed as D
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.side_adapter import SideAdapter
    rank, world, _ = D.init_from_env("gloo")
    torch.cuda.set_device(0)
    arch = dict(width=256, layers=4, heads=4, patch=16, resolution=64, embed_dim=64)
    K, T = 7, 7
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_shard").set(thing_classes=names)
    which = os.environ.get("OVIS_SHARD_ARCH", "brivis")          # brivis | san_online | openvis_online: the three frame-shardable architectures
    cfg = config.get_cfg()
    cfg.MODEL.PRECISION = "fp32"
    if os.environ.get("OVIS_SHARD_WINDOWS") == "1":              # the rank's own frames as windows of 2 (minvis.py:340-362) in front of the exchange
        cfg.MODEL.MASK_FORMER.TEST.WINDOW_INFERENCE, cfg.MODEL.MASK_FORMER.TEST.WINDOW_SIZE = True, 2
    if which == "openvis_online":
        cfg.MODEL.META_ARCHITECTURE = "OpenVISOnline"
        cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "FrameMultiScaleMaskedTransformerDecoder"
        cfg.MODEL.CLIP_ADAPTER.PRECISION = "fp32"
        model = config.build_model(cfg)
        model.load_state_dict(weights.random_init(weights.openvis_spec("r50", None, 100), seed=5))
        model.clip_adapter.set_text_features(names, bench.synth_text(K, 512, spread=0.25))
    else:
        cfg.MODEL.META_ARCHITECTURE = "BriVIS" if which == "brivis" else "SANOnline"
        cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "SideAdapterFrameMultiScaleMaskedTransformerDecoder"
        cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS = 4
        model = config.build_model(cfg)
        model.clip_adapter = SideAdapter("tiny", broken_idx=3, merge_ids=[1, 2, 3], num_queries=100, arch=arch, precision="fp32")
        spec = weights.brivis_spec("r50", arch, 100) if which == "brivis" else weights.san_spec("r50", arch, 100)
        model.load_state_dict(weights.random_init(spec, seed=5))
        model.clip_adapter.set_text_features(names, bench.synth_text(K, 64))
    frames = bench.synth_frames(T, 90, 120, 3, "cpu")
    inp = [{"image": [f for f in frames], "dataset_name": "synthetic_shard"}]
    fr = D.inference_shard(T, rank, world)
    st = {}
    gather_to = int(os.environ["OVIS_GATHER_TO"]) if "OVIS_GATHER_TO" in os.environ else None
    out = model(inp, stages=st, frame_range=(fr.start, fr.stop) if world > 1 else None, gather_masks_to=gather_to)
    res = {"rank": rank, "world": world, "range": [fr.start, fr.stop], "labels": out["pred_labels"], "scores": out["pred_scores"],
           "queries": out["pred_queries"], "indices": st["indices"].cpu().tolist(), "probs": st["probs"].clamp(min=0).cpu().tolist(),
           "mask_sums": [int(m.sum()) for m in out["pred_masks"]],
           "mask_shape": list(out["pred_masks"][0].shape) if len(out["pred_masks"]) else [],
           "frame_sums": [int(v) for v in torch.stack(list(out["pred_masks"])).sum(dim=(0, 2, 3))] if len(out["pred_masks"]) else [],
           "mask_frames": list(out.get("pred_masks_frames", []))}
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    D.barrier()


if __name__ == "__main__":
    main()
