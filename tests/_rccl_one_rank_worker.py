"""Worker of tests/test_sharded_gpu.py::test_one_rank_rccl_group_runs_the_frame_sharded_path: ONE rank in a real "nccl" (= RCCL) process
group on the box's single GPU.  A 1-rank group is legal, and it is the only way to execute the RCCL branches of
openvis_amd/distributed.py on a 1-GPU box: `all_gather_into_tensor(async_op=True)` on the side stream + `work.wait()`, the device
`all_reduce`, `dist.gather` of device tensors -- so that the first 8-GPU run is not the first run of that code.

Runs (1) the primitive checks of tests/_rccl_worker.py at world 1, (2) BriVIS with frame_range=(0, T) + gather_masks_to=0 through
all_gather_frames_async / all_reduce_sum / gather_frame_masks against the un-sharded forward of the same model (equal outputs), (3) an
out-of-memory error injected BEHIND the all-gather: the frame-sharded forward must re-raise at once, with exactly one all-gather issued
(a local retry would issue a second one while the peers have moved on).  The process group is created before anything else
touches the GPU."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29677")
    torch.cuda.set_device(0)
    from openvis_amd import distributed as D
    rank, world, _ = D.init_from_env("nccl", force=True)
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", 0)
    D.warm_up(dev)

    # (1) the primitives, on device tensors through RCCL
    T, Q, C = 7, 100, 256
    full = torch.arange(T * Q * C, dtype=torch.float32).view(T, Q, C) * 1e-3
    h = D.all_gather_frames_async(full.to(dev), T)
    assert h.work is not None and h.side is not None, "the RCCL branch (side stream + async work) was not taken"
    busy = torch.ones(512, 512, device=dev) @ torch.ones(512, 512, device=dev)
    got = h.wait()
    assert got.is_cuda and torch.equal(got.cpu(), full) and float(busy[0, 0]) == 512.0
    s = D.all_reduce_sum(torch.full((Q, 483), 3.0, device=dev))
    assert s.is_cuda and torch.equal(s.cpu(), torch.full((Q, 483), 3.0))
    masks = (torch.arange(10 * T * 8 * 16, device=dev) % 2).to(torch.uint8).view(10, T, 8, 16)
    g = D.gather_frame_masks(masks, T, dst=0)
    assert g.is_cuda and torch.equal(g, masks)
    assert D.max_over_ranks(2.5, dev) == 2.5
    blk = torch.arange(27300, dtype=torch.float32, device=dev)       # the split-KV decoder's per-layer exchange (one packed flash partial)
    rows = D.all_gather_rows(blk)
    assert rows.is_cuda and rows.shape == (1, 27300) and torch.equal(rows[0], blk) and rows.data_ptr() != blk.data_ptr()

    # (2) BriVIS frame-"sharded" over the 1-rank RCCL group == the un-sharded forward
    import bench
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.side_adapter import SideAdapter
    arch = dict(width=256, layers=4, heads=4, patch=16, resolution=64, embed_dim=64)
    K, T = 7, 6
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_shard").set(thing_classes=names)
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "BriVIS"
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "SideAdapterFrameMultiScaleMaskedTransformerDecoder"
    cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS = 4
    model = config.build_model(cfg)                                  # default policy (mixed, fp16x2): the range flag rides on the all-reduce
    model.clip_adapter = SideAdapter("tiny", broken_idx=3, merge_ids=[1, 2, 3], num_queries=100, arch=arch, precision="fp32")
    model.load_state_dict(weights.random_init(weights.brivis_spec("r50", arch, 100), seed=5))
    model.clip_adapter.set_text_features(names, bench.synth_text(K, 64))
    frames = bench.synth_frames(T, 90, 120, 3, "cpu")
    inp = [{"image": [f for f in frames], "dataset_name": "synthetic_shard"}]
    st0, st1 = {}, {}
    plain = model(inp, stages=st0)
    calls = {"gather": 0, "reduce": 0, "masks": 0}
    real_g, real_r, real_m = D.all_gather_frames_async, D.all_reduce_sum, D.gather_frame_masks
    D.all_gather_frames_async = lambda *a, **k: (calls.__setitem__("gather", calls["gather"] + 1), real_g(*a, **k))[1]
    D.all_reduce_sum = lambda *a, **k: (calls.__setitem__("reduce", calls["reduce"] + 1), real_r(*a, **k))[1]
    D.gather_frame_masks = lambda *a, **k: (calls.__setitem__("masks", calls["masks"] + 1), real_m(*a, **k))[1]
    sharded = model(inp, stages=st1, frame_range=(0, T), gather_masks_to=0)
    assert calls == {"gather": 1, "reduce": 1, "masks": 1}, calls
    assert torch.equal(st0["indices"].cpu(), st1["indices"].cpu())
    dp = (st0["probs"] - st1["probs"]).abs().max().item()
    assert dp < 1e-5, dp
    assert plain["pred_labels"] == sharded["pred_labels"] and plain["pred_queries"] == sharded["pred_queries"]
    assert max(abs(a - b) for a, b in zip(plain["pred_scores"], sharded["pred_scores"])) < 1e-5
    pm, sm = torch.stack(list(plain["pred_masks"])), torch.stack(list(sharded["pred_masks"]))
    assert pm.shape == sm.shape == (10, T, 90, 120) and torch.equal(pm.cpu(), sm.cpu())

    # (3) an out-of-memory error behind the all-gather: re-raised at once, one all-gather issued
    calls.update(gather=0, reduce=0, masks=0)
    real_post = model.clip_adapter.post_encode_image
    n_post = []

    def oom(*a, **k):
        n_post.append(1)
        raise torch.OutOfMemoryError("HIP out of memory (injected by the test)")
    model.clip_adapter.post_encode_image = oom
    try:
        raised = False
        try:
            model(inp, frame_range=(0, T), gather_masks_to=0)
        except torch.OutOfMemoryError:
            raised = True
        assert raised and len(n_post) == 1 and calls["gather"] == 1 and calls["reduce"] == 0, (raised, n_post, calls)
        # the un-sharded forward of the same model still takes the ladder (run, run again, then windows)
        n_post.clear()
        raised = False
        try:
            model(inp)
        except torch.OutOfMemoryError:
            raised = True
        assert raised and len(n_post) >= 2, n_post
    finally:
        model.clip_adapter.post_encode_image = real_post
        D.all_gather_frames_async, D.all_reduce_sum, D.gather_frame_masks = real_g, real_r, real_m
    torch.cuda.synchronize()
    rccl = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
    json.dump({"ok": True, "backend": dist.get_backend(), "world": dist.get_world_size(), "librccl": rccl, "probs_diff": dp}, open(out_path, "w"))
    dist.destroy_process_group()
    print("RCCL1_OK", rccl)


if __name__ == "__main__":
    main()
