#!/usr/bin/env python3
"""Thin eval driver with the reference's command line (train_net.py:303-313: --config-file / --eval-only / --num-gpus /
KEY VALUE opts) for the MI355X-native hot path.

    python train_net.py --config-file configs/openvoc_ytvis_coco/openvis_R50_bs16_6000st.yaml --eval-only \
        --num-gpus 8 --input /data/videos --classes classes.txt --output results.json MODEL.WEIGHTS model_final.pth

Only what surrounds `model.forward` at eval time lives here (datasets, mappers and evaluators are out of
scope): videos are directories of frame images under --input (or --synthetic N clips), frames are resized on the GPU exactly like
the test-time mapper (shortest edge = INPUT.MIN_SIZE_TEST, PIL bilinear), videos are sharded over the ranks in the InferenceSampler layout
(data/build.py:238-247), results are gathered on rank 0 and written as the result list the reference's evaluator dumps —
YTVIS tracks (ytvis_eval.py:258-301) or, for a `burst_*` test set, BURST sequences (burst_eval.py:177-240) — with
GPU-encoded COCO RLE segmentations.  Training is not part of this tier."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_video_dir(path, min_size, max_size, device):
    """decode on the host (PIL), ResizeShortestEdge + BILINEAR on the GPU, bit-exact with the reference's mapper
    (ytvis_dataset_mapper.py:298-313 -> openvis_amd/data.py)."""
    from PIL import Image
    from openvis_amd import data
    names = sorted(n for n in os.listdir(path) if n.lower().endswith((".jpg", ".jpeg", ".png")))
    decoded = [np.asarray(Image.open(os.path.join(path, n)).convert("RGB")) for n in names]
    return data.load_and_resize(decoded, min_size, max_size, device) + (names,)


def worker(rank, world, args):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(args.port))
    torch.cuda.set_device(rank)
    from openvis_amd import config, evals, weights, distributed as D
    from openvis_amd.catalog import MetadataCatalog
    D.init_from_env("nccl")
    cfg = config.get_cfg()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.MODEL.DEVICE = f"cuda:{rank}"
    cfg.MODEL.MASK_FORMER.TEST.OUTPUT_RLE = True
    model = config.build_model(cfg)
    model.device = torch.device("cuda", rank)
    if cfg.MODEL.WEIGHTS and os.path.isfile(cfg.MODEL.WEIGHTS):
        sd = weights.load_checkpoint(cfg.MODEL.WEIGHTS)
    else:
        if rank == 0:
            print(f"[train_net] MODEL.WEIGHTS '{cfg.MODEL.WEIGHTS}' not found: seeded random-init weights of the configured architecture")
        sd = weights.random_init(weights.spec_for_cfg(cfg), seed=cfg.SEED)
    model.load_state_dict(sd)
    if args.classes:
        names = [l.strip() for l in open(args.classes) if l.strip()]
    else:
        names = [f"class_{i}" for i in range(40)]
    dataset = cfg.DATASETS.TEST[0]
    MetadataCatalog.get(dataset).set(thing_classes=names)
    ad = model.clip_adapter
    if getattr(ad, "text_tower", None) is None or args.synthetic_text:
        g = torch.Generator().manual_seed(1)
        dim = ad.arch["embed_dim"]
        base = torch.randn(1, dim, generator=g)
        clean = [ad._clean(n) for n in names] if hasattr(ad, "_clean") else names
        ad.set_text_features(clean, torch.nn.functional.normalize(base + 0.05 * torch.randn(len(names), dim, generator=g), dim=-1))
    if args.synthetic:
        import bench
        videos = [(f"synthetic_{i}", None) for i in range(args.synthetic)]
    else:
        videos = [(d, os.path.join(args.input, d)) for d in sorted(os.listdir(args.input)) if os.path.isdir(os.path.join(args.input, d))]
    results = []

    def load(vi):
        vid, path = videos[vi]
        if path is None:
            frames = [f for f in bench.synth_frames(args.frames, 360, 640, 1000 + vi, "cpu")]
            hw, fnames = (360, 640), [f"frame{i:04d}.jpg" for i in range(args.frames)]
        else:
            frames, hw, fnames = load_video_dir(path, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.get("MAX_SIZE_TEST", 1333), model.device)
        return [{"image": frames, "dataset_name": dataset, "height": hw[0], "width": hw[1], "video_id": vid, "length": len(frames),
                 "seq_name": vid, "dataset": os.path.basename(os.path.normpath(args.input)) if args.input else "synthetic",
                 "annotated_image_paths": fnames}]

    def emit(inputs, out):
        if not out["pred_scores"]:
            out["pred_masks_rle"] = []
        # the evaluators then un-map contiguous ids to dataset ids (ytvis_eval.py:152-168, burst_eval.py:145-158); without
        # dataset metadata the mapping is the 1-based identity
        if dataset.startswith("burst"):
            for seq in evals.instances_to_burst_json_video(inputs, out):
                seq["track_category_ids"] = {k: l + 1 for k, l in seq["track_category_ids"].items()}
                results.append(seq)
        else:
            for r in evals.instances_to_coco_json_video(inputs, out):
                r["category_id"] += 1
                results.append(r)

    shard = list(D.inference_shard(len(videos), rank, world))
    if args.streams > 1:
        # several videos in flight on this GPU (openvis_amd/runtime.py): chunks bound what is decoded ahead
        from openvis_amd.runtime import ClipPipeline
        pipe = ClipPipeline(model, args.streams)
        for c0 in range(0, len(shard), 4 * args.streams):
            chunk = [load(vi) for vi in shard[c0:c0 + 4 * args.streams]]
            for inputs, out in zip(chunk, pipe.run(chunk)):
                emit(inputs, out)
    else:
        for vi in shard:
            inputs = load(vi)
            emit(inputs, model(inputs))
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(results, gathered, dst=0)                  # comm.gather of the evaluator (ytvis_eval.py:122-128)
        results = [r for part in gathered for r in part] if rank == 0 else []
    if rank == 0:
        json.dump({"sequences": results} if dataset.startswith("burst") else results, open(args.output, "w"))   # burst_eval.py:160
        kind = "BURST sequences" if dataset.startswith("burst") else "instance tracks"
        print(f"[train_net] {len(videos)} videos, {len(results)} {kind} -> {args.output}")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--input", default="", help="directory of videos (one sub-directory of frame images per video)")
    ap.add_argument("--synthetic", type=int, default=0, help="run N synthetic clips instead of --input")
    ap.add_argument("--frames", type=int, default=5, help="frames per synthetic clip")
    ap.add_argument("--classes", default="", help="text file with one class name per line (default: 40 placeholder names)")
    ap.add_argument("--synthetic-text", action="store_true", help="random unit text embeddings even if the checkpoint has a text tower")
    ap.add_argument("--output", default="results.json")
    ap.add_argument("--port", type=int, default=29511)
    ap.add_argument("--streams", type=int, default=1, help="videos in flight per GPU (HIP streams, openvis_amd/runtime.py)")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    args = ap.parse_args()
    if not args.eval_only:
        sys.exit("train_net.py: only --eval-only is implemented (training is outside the hot-path tier, SURVEY.md 8)")
    if not args.input and not args.synthetic:
        sys.exit("train_net.py: give --input DIR or --synthetic N")
    if args.num_gpus > 1:
        import torch.multiprocessing as mp
        mp.spawn(worker, args=(args.num_gpus, args), nprocs=args.num_gpus)
    else:
        worker(0, 1, args)


if __name__ == "__main__":
    main()
