"""bench.py — OpenVIS R50 720p eval-only throughput on MI355X (BASELINE.json metric, configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; ranks process independent clips — the OpenVIS
   offline decoder attends over all frames of a clip, so the path shards by clip with no data-path collective,
   SURVEY.md §8(e); "scaling": "weak".)

A "step" is one full eval forward of the OpenVIS meta-architecture over one synthetic 5-frame 720p clip: `model(batched_inputs)` wall-clock with
the clip's uint8 frames RESIDENT IN HBM when the timed region starts (the task's measurement rule; `host_inputs` on the same line is the same
loop with the clips in pinned host memory, i.e. including the 13.8 MB upload per step) and INCLUDING the D2H copy of the 10
output masks: pre-process -> ResNet-50 -> MSDeformAttn pixel decoder -> 9-layer
masked-attention decoder -> mask boxes -> CLIP ViT-B/16 on every valid (frame, query) crop -> class aggregation ->
top-10 -> output masks copied to the host.  Random-init weights of the real architecture (no network for
checkpoints), 482 synthetic class embeddings (burst_val size).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

T_CLIP, H720, W720, NUM_CLASSES, NUM_QUERIES = 5, 720, 1280, 482, 100
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_F16_MFMA_TFLOPS = 2500.0     # same guide: Peak BF16/FP16 MFMA, dense (~2.5 PF)
PEAK_HBM_GBS = 8000.0


def csrc_digest():
    """sha256 over the kernel sources (openvis_amd/csrc: *.hip, *.h, *.cpp, Makefile, torch_ext/*; include/*.h), in sorted path order.  tools/pmc_traffic.py
    stores it with the PMC traffic summary; the bench line only quotes a summary whose digest matches the sources it runs on."""
    import glob
    import hashlib
    base = os.path.join(ROOT, "openvis_amd", "csrc")
    files = sorted(f for pat in ("*.hip", "*.h", "*.cpp", "Makefile", "torch_ext/*.cpp", "torch_ext/*.h", "../../include/*.h") for f in glob.glob(os.path.join(base, pat)))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, base).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def synth_frames(T, H, W, seed, device):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(T, 3, H, W, generator=g) * 255
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    for t in range(T):
        for (cy, cx, sy, sx, amp) in ((0.35, 0.3, 0.12, 0.1, 110), (0.6, 0.65, 0.2, 0.15, 90), (0.5, 0.5, 0.4, 0.45, 40)):
            base[t] = base[t] * 0.7 + amp * torch.exp(-(((yy - (cy + 0.01 * t) * H) / (sy * H)) ** 2 +
                                                        ((xx - (cx - 0.015 * t) * W) / (sx * W)) ** 2))
    return base.clamp(0, 255).to(torch.uint8).to(device)


def synth_text(K, dim, seed=1, spread=0.05):
    """Unit rows [K, dim]: one common direction + `spread` x independent noise.  0.05 (the timing default; timing does not depend on
    it) gives x100 cosine logits that differ by ~0.2 between classes: a near-uniform softmax whose top-k is decided by near-ties.
    The parity tests use spread = 0.25 -- logits spread by ~1, separated classes, an un-saturated softmax -- so that the top-10
    (query, label) set and its scores are a sharp statement (tests/_logits.py:check_top10)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, dim, generator=g)
    return torch.nn.functional.normalize(base + spread * torch.randn(K, dim, generator=g), dim=-1)


def _aggregate(prof):
    """ops.PROFILE entries -> ({kernel: [launches, flops, seconds]}, {kernel: [launches, bytes, seconds]}) (HIP events per launch)."""
    agg, hbm = {}, {}
    for name, work, e0, e1 in prof:
        a = (hbm if name.startswith("hbm:") else agg).setdefault(name.replace("hbm:", ""), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += work
        a[2] += e0.elapsed_time(e1) * 1e-3
    return agg, hbm


def _family_of(k):
    # the ping-pong kernel's f32-A instantiations (<.., X3=false, FA=true>: bf16x2, three bf16 products per f32 product) are a
    # family of their own: their flops are f32-equivalent and their ceiling is the bf16 peak / 3, not the fp16 peak
    base = k.split("<")[0]
    targs = k[len(base) + 1:-1].split(",") if "<" in k else []          # <OUT, ACT, HAS_R, X3, FA, R16>
    return base + "[f32A]" if base == "gemm_f16_pp_kernel" and len(targs) >= 5 and targs[4] == "true" else base


def _families(a):
    """"The dominant kernel" = the kernel TEMPLATE with the largest summed launch time (rocprofv3 lists every instantiation as its own row
    -- <out dtype, activation, residual> for the fp16 GEMM --, but it is one piece of code)."""
    fam = {}
    for k, v in a.items():
        f = fam.setdefault(_family_of(k), [0, 0.0, 0.0, {}])
        f[0] += v[0]; f[1] += v[1]; f[2] += v[2]; f[3][k] = v
    return fam


def _family_peak(kbase, f32_split):
    """(peak TFLOP/s, note) of a GEMM kernel family: one f32 product = 6 (bf16x3) or 3 (bf16x2 / fp16x2) 16-bit MFMA products."""
    if "f32x3" in kbase or "[f32A]" in kbase:
        nprod = 3 if (f32_split in ("bf16x2", "fp16x2") or "[f32A]" in kbase) else 6
        return round(PEAK_F16_MFMA_TFLOPS / nprod, 1), (f"{'fp16' if f32_split == 'fp16x2' else 'bf16'} dense MFMA peak / {nprod} "
                                                         f"({f32_split}: {nprod} 16-bit products per f32 product)")
    if "f16" in kbase:
        return PEAK_F16_MFMA_TFLOPS, "fp16 dense MFMA peak"
    return PEAK_F32_MFMA_TFLOPS, "f32 MFMA peak"


def _family_label(kbase, members):
    return kbase.replace("[f32A]", "") + ("<*,FA=true>" if "[f32A]" in kbase else "<*>") if len(members) > 1 else next(iter(members))


def measure_other_config(name, device, args, f32_split):
    """One of BASELINE.json's other single-GPU-runnable configs (configs[2] san_online, [3] brivis 36 frames, [4] brivis Swin-L 1080p) on the
    DEFAULT command's line: built, warmed, timed like the headline (barrier-free: N = 1), then one extra pass under per-launch HIP events
    for the roofline of ITS dominant GEMM family.  The model is dropped afterwards (weights + activations of Swin-L / ViT-L: ~6 GB)."""
    from openvis_amd import ops
    T = 36 if name.startswith("brivis") else T_CLIP
    res = 1080 if name.endswith("_swinl") else 720
    FH, FW = res, res * 16 // 9
    steps = {"san_online": 20, "brivis": 6, "brivis_swinl": 2}.get(name, 4)
    m, _, _ = build_model(device, clip_precision=args.clip_precision, precision=args.precision, model_name=name, f32_split=args.f32_split)
    clips = [synth_frames(T, FH, FW, 1000 + i, "cpu").to(device) for i in range(2)]           # resident in HBM, like the headline's
    inputs = [[{"image": [f for f in c], "dataset_name": "synthetic_burst_val"}] for c in clips]
    out = None
    for i in range(2):
        out = m(inputs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = m(inputs[i % 2])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if out is not None and hasattr(out, "wait"):
        out.wait()
    ops.PROFILE = []
    m(inputs[0])
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg, hbm = _aggregate(prof)
    fam = _families(agg)
    kbase, (n_launch, flops, secs, members) = max(fam.items(), key=lambda kv: kv[1][2])
    peak, peak_note = _family_peak(kbase, f32_split)
    roof = {"kernel": _family_label(kbase, members), "bound": "mfma", "achieved": round(flops / secs / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(flops / secs / 1e12 / peak, 4), "peak_note": peak_note, "launches_per_step": n_launch,
            "ms_per_step_in_kernel": round(secs * 1e3, 3)}
    k1 = None
    if hbm:
        kname, (kn, kbytes, ksecs) = max(hbm.items(), key=lambda kv: kv[1][2])
        k1 = {"kernel": kname, "bound": "hbm", "achieved": round(kbytes / ksecs / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "frac": round(kbytes / ksecs / 1e9 / PEAK_HBM_GBS, 4), "launches_per_step": kn}
    bb = {"r50": "R50", "swin_l": "Swin-L"}[MODELS[name].get("backbone", "r50")]
    line = {"workload": f"{name} {bb} {res}p, {T}-frame clips, {MODELS[name].get('clip', 'ViT-B/16')}, 100 queries, 482 classes",
            "baseline_config": {"san_online": "configs[2]", "brivis": "configs[3] (one GPU: all 36 frames on this rank)",
                                "brivis_swinl": "configs[4] (one GPU: all 36 frames on this rank)"}.get(name),
            "value": round(T * steps / elapsed, 3), "unit": "frames/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": 2,
            "frames_per_step": T, "dtype": "f16" if "fp16" in (m.backbone.precision, m.clip_adapter.precision) else "f32 (fp16x2 on the 16-bit MFMA)"
            if f32_split == "fp16x2" else "f32", "roofline": roof, "roofline_k1": k1}
    del m, out, clips, inputs
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return line


MODELS = {   # --model: META_ARCHITECTURE, decoder, weight spec, backbone, CLIP tower, queries
    "openvis": dict(arch="OpenVIS", decoder="VideoMultiScaleMaskedTransformerDecoder", spec="openvis_spec"),
    "openvis_online": dict(arch="OpenVISOnline", decoder="FrameMultiScaleMaskedTransformerDecoder", spec="openvis_spec"),
    "san_online": dict(arch="SANOnline", decoder="SideAdapterFrameMultiScaleMaskedTransformerDecoder", spec="san_spec"),
    "brivis": dict(arch="BriVIS", decoder="SideAdapterFrameMultiScaleMaskedTransformerDecoder", spec="brivis_spec"),
    # BASELINE.json configs[4]: Swin-L (swin/openvis_swinL_*.yaml:5-9) + CLIP ViT-L/14@336 side adapter
    # (swin/brivis_SwinB_*.yaml:17-22: MERGE_IDS [6,12,18], BROKEN_ID 21, 16 heads, embed 768)
    "brivis_swinl": dict(arch="BriVIS", decoder="SideAdapterFrameMultiScaleMaskedTransformerDecoder", spec="brivis_spec",
                         backbone="swin_l", clip="ViT-L/14@336px"),
    "openvis_swinl": dict(arch="OpenVIS", decoder="VideoMultiScaleMaskedTransformerDecoder", spec="openvis_spec",
                          backbone="swin_l", clip="ViT-L/14@336px", queries=200),
}


def build_model(device, seed=42, clip_precision="fp16", precision="mixed", model_name="openvis", f32_split="auto", crop_list="auto"):
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.adapter import _CLIP_ARCH
    m = MODELS[model_name]
    backbone, clip, queries = m.get("backbone", "r50"), m.get("clip", "ViT-B/16"), m.get("queries", 100)
    cfg = config.get_cfg()
    cfg.MODEL.DEVICE = str(device)
    cfg.MODEL.META_ARCHITECTURE = m["arch"]
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = m["decoder"]
    cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES = queries
    cfg.MODEL.CLIP_ADAPTER.PRECISION = clip_precision
    cfg.MODEL.CLIP_ADAPTER.CLIP_MODEL_NAME = clip
    if clip == "ViT-L/14@336px":
        cfg.MODEL.CLIP_ADAPTER.MERGE_IDS, cfg.MODEL.CLIP_ADAPTER.BROKEN_ID = [6, 12, 18], 21
        cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS, cfg.MODEL.CLIP_ADAPTER.CLIP_EMBED_DIMS = 16, 768
    if backbone != "r50":
        a = weights.SWIN_ARCH[backbone]
        cfg.MODEL.BACKBONE.NAME = "D2SwinTransformer"
        cfg.MODEL.SWIN.EMBED_DIM, cfg.MODEL.SWIN.DEPTHS = a["embed_dim"], list(a["depths"])
        cfg.MODEL.SWIN.NUM_HEADS, cfg.MODEL.SWIN.WINDOW_SIZE = list(a["num_heads"]), a["window"]
    cfg.MODEL.PRECISION = precision
    cfg.MODEL.F32_GEMM_SPLIT = f32_split
    cfg.MODEL.CLIP_ADAPTER.CROP_LIST = crop_list
    model = config.build_model(cfg)
    model.device = torch.device(device)
    sd = weights.random_init(getattr(weights, m["spec"])(backbone, _CLIP_ARCH[clip], queries), seed=seed)
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(NUM_CLASSES)]
    MetadataCatalog.get("synthetic_burst_val").set(thing_classes=names)
    text = synth_text(NUM_CLASSES, _CLIP_ARCH[clip]["embed_dim"])
    model.clip_adapter.set_text_features(names, text)
    return model, sd, text


def _usable_cpus(cap=32):
    """CPU threads for the oracle leg: affinity mask, cgroup quota and a cap of 32 (beyond that the oracle's small
    torch ops stop scaling; a 256-thread run on the GPU box's host was ~50x SLOWER than this)."""
    n = torch.get_num_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(sd, iters=3):
    """BASELINE.md section 3 / SURVEY.md 8(d): the oracle (CPU port of the reference path, oracle/torch_ref.py, fp32) on C1 =
    BASELINE.json configs[0]: ONE 480x854 frame (seed 0), 100 queries, K = 40 classes, the COMPLETE forward (nothing
    extrapolated): 1 warm-up + `iters` timed iterations, median s/frame, per-stage split of the median iteration."""
    from oracle import torch_ref as TR
    import torch.nn.functional as F
    cores = _usable_cpus()
    torch.set_num_threads(cores)
    frames = synth_frames(1, 480, 854, 0, "cpu")
    text = synth_text(40, 512)
    pc = time.perf_counter

    def one():
        t = {}
        with torch.no_grad():
            t0 = pc()
            images, (H, W) = TR.preprocess([f for f in frames])
            feats = TR.resnet50(images, sd)
            t["backbone"] = pc() - t0; t0 = pc()
            mask_features, _, ms = TR.pixel_decoder(feats, sd)
            t["pixel_decoder"] = pc() - t0; t0 = pc()
            _, pred_masks = TR.video_decoder(ms, mask_features, sd)
            t["decoder"] = pc() - t0; t0 = pc()
            mask_pred = F.interpolate(pred_masks[0], size=images.shape[-2:], mode="bilinear", align_corners=False)
            t["mask_upsample"] = pc() - t0; t0 = pc()
            probs, vmasks, extras = TR.open_vocabulary_inference(mask_pred, frames, text, sd)
            t["clip_crops_logits"] = pc() - t0; t0 = pc()
            TR.inference_video(pred_masks.shape[1], text.shape[0], probs, vmasks, (H, W), H, W)
            t["topk_output_masks"] = pc() - t0
        t["n_valid_crops"] = int(extras["valid"].sum()) if "valid" in extras else -1
        return t

    one()                                                             # warm-up (allocator, thread pools)
    runs = [one() for _ in range(iters)]
    tot = lambda r: sum(v for k, v in r.items() if k != "n_valid_crops")
    runs.sort(key=tot)
    med = runs[len(runs) // 2]
    return {"value": round(1.0 / tot(med), 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "os_cpu_count": os.cpu_count(), "torch_num_threads": torch.get_num_threads(),
            "s_per_frame_median": round(tot(med), 3), "s_per_frame_all": [round(tot(r), 3) for r in runs],
            "stage_s": {k: round(v, 3) for k, v in med.items() if k != "n_valid_crops"},
            "sample": f"oracle/torch_ref.py fp32, C1 = BASELINE.json configs[0]: one 480x854 frame, 100 queries, 40 classes, "
                      f"complete eval forward ({med['n_valid_crops']} valid crops through CLIP ViT-B/16), 1 warm-up + {iters} "
                      f"timed iterations, median; threads = min(torch default, affinity, cgroup quota, 32)"}


def collective_spans(model_fn, inputs, steps=3):
    """Frame-sharded runs: what every rank spends in the exchange steps, `steps` untimed extra steps.  host=True spans are
    host wall time around the call (the wait for the side-stream all-gather, the all-reduce, the mask gather: what the rank loses to the
    collective including the wait for the slowest rank); the others are HIP-event times of replicated work."""
    from openvis_amd import distributed as D
    D.SPANS = {}
    for i in range(steps):
        o_ = model_fn(inputs[i % len(inputs)])
        if hasattr(o_, "wait"):
            o_.wait()
    torch.cuda.synchronize()
    mine = D.spans_ms()
    D.SPANS = None
    per_rank = D.gather_objects(mine)
    return {"per_rank": per_rank, "max_over_ranks": {k: max(r.get(k, 0.0) for r in per_rank) for k in per_rank[0]},
            "note": f"mean of {steps} untimed steps; all_gather_wait / logit_all_reduce / logit_all_gather / mask_gather = host wall time around the call, "
                    "linker / temporal_resampler / partial_all_gather (one of the nine per clip; RCCL) = HIP-event time on the compute stream"}


def measure_frame_sharded(device, rank, world, rig, args, sync_all):
    """north_star's multi-GPU split, measured by the DEFAULT multi-rank command: ONE BriVIS R50 clip of `--sharded-frames` 720p frames
    (BASELINE.json configs[3]: 36), contiguous frame blocks per rank (36 over 8 ranks: 5,5,5,5,4,4,4,4), one RCCL all-gather of the
    per-frame query embeddings in front of the replicated Brownian-bridge linker + temporal resampler, an all-reduce of the per-frame logit
    sums (reference exchange point: modeling/minvis.py:28-72, brivis.py:173-176).  Runs in the process group of the
    headline, after it; strong scaling: value = frames of the clip / max-over-ranks step time."""
    from openvis_amd import distributed as D
    T = args.sharded_frames
    model, _, _ = build_model(device, clip_precision=args.clip_precision, precision=args.precision, model_name="brivis", f32_split=args.f32_split)
    fr = D.inference_shard(T, rank, world)
    kw = {"frame_range": (fr.start, fr.stop)}
    if args.gather_masks:
        kw["gather_masks_to"] = 0
    clips = [synth_frames(T, H720, W720, 1000 + i, "cpu").to(device) if args.inputs == "device" else synth_frames(T, H720, W720, 1000 + i, "cpu").pin_memory()
             for i in range(2)]
    inputs = [[{"image": [f for f in c], "dataset_name": "synthetic_burst_val"}] for c in clips]
    fn = lambda inp: model(inp, **kw)
    D.warm_up(device if not rig else "cpu")                  # the first all-gather / all-reduce / gather of the communicator: untimed
    n = max(2, min(20, args.steps // 5))
    for i in range(2):
        fn(inputs[i % 2])
    sync_all()
    t0 = time.perf_counter()
    out = None
    for i in range(n):
        out = fn(inputs[i % 2])
    sync_all()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, "cpu" if rig else device)
    if out is not None and hasattr(out, "wait"):
        out.wait()
    spans = collective_spans(fn, inputs)
    return {"value": round(T * n / elapsed, 3), "unit": "frames/s", "ms_per_step": round(elapsed / n * 1e3, 3), "steps": n, "scaling": "strong",
            "frames_per_rank": [len(D.inference_shard(T, r, world)) for r in range(world)], "world_size_seen": D.world_size(),
            "collective_ms": spans,
            "workload": f"brivis R50 720p, ONE {T}-frame clip frame-sharded x{world}, RCCL all-gather of [t,100,256] query embeddings + "
                        f"replicated linker / resampler + logit all-reduce" + (" + mask gather to rank 0" if args.gather_masks else ""),
            "note": "BASELINE.json configs[3] (north_star's split) in the same process group as the clip-replica headline; not the headline"}


def measure_split_clip(model, device, rank, world, rig, args, sync_all):
    """The OpenVIS row of SURVEY.md 8(e): ONE clip of `--split-frames` 720p frames over the ranks -- backbone, pixel decoder and CLIP crops on the
    rank's own frames, the offline decoder's cross-attention (joint over all frames: video decoder:397-403, 417-426) as split-KV: per layer
    every rank's flash partial over its own keys (109 KB) is all-gathered over RCCL and merged; the crop logits are all-gathered once for the
    per-query mean.  Strong scaling: value = frames of the clip / max-over-ranks step time.  Rank 0 then runs the same clip un-split, alone,
    as the in-run one-GPU reference.  Uses the headline's model (same weights, same policy)."""
    from openvis_amd import distributed as D
    T = max(args.split_frames, world)
    fr = D.inference_shard(T, rank, world)
    kw = {"frame_range": (fr.start, fr.stop)}
    if args.gather_masks:
        kw["gather_masks_to"] = 0
    clips = [synth_frames(T, H720, W720, 2000 + i, "cpu").to(device) if args.inputs == "device" else synth_frames(T, H720, W720, 2000 + i, "cpu").pin_memory()
             for i in range(2)]
    inputs = [[{"image": [f for f in c], "dataset_name": "synthetic_burst_val"}] for c in clips]
    fn = lambda inp: model(inp, **kw)
    D.warm_up(device if not rig else "cpu")
    if os.environ.get("OVIS_BENCH_FAIL_SIDE_RANK") == str(rank):       # tests: this rank leaves the collective sequence here
        raise RuntimeError("injected by OVIS_BENCH_FAIL_SIDE_RANK")
    n = max(2, min(20, args.steps // 5))
    for i in range(2):
        fn(inputs[i % 2])
    sync_all()
    t0 = time.perf_counter()
    out = None
    for i in range(n):
        out = fn(inputs[i % 2])
    sync_all()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, "cpu" if rig else device)
    if out is not None and hasattr(out, "wait"):
        out.wait()
    spans = collective_spans(fn, inputs)
    alone = None
    if rank == 0:                                            # the other ranks wait at sync_all's barrier
        for i in range(2):
            model(inputs[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            o_ = model(inputs[i % 2])
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t0) / n
        if hasattr(o_, "wait"):
            o_.wait()
    sync_all()
    return {"value": round(T * n / elapsed, 3), "unit": "frames/s", "ms_per_step": round(elapsed / n * 1e3, 3), "steps": n, "scaling": "strong",
            "frames_per_rank": [len(D.inference_shard(T, r, world)) for r in range(world)], "world_size_seen": D.world_size(),
            "unsplit_on_one_gpu": ({"ms_per_step": round(alone * 1e3, 3), "value": round(T / alone, 3), "unit": "frames/s"} if alone else None),
            "collective_ms": spans,
            "workload": f"openvis R50 720p, ONE {T}-frame clip over {world} rank(s): per-frame stages local, 9 x all-gather of the decoder's "
                        f"[100 x 8 x (32 + 2) + 100] f32 flash partials + 1 x all-gather of the [t,100,483] crop logits"
                        + (" + mask gather to rank 0" if args.gather_masks else ""),
            "note": "SURVEY.md 8(e), OpenVIS row (optional split-KV) in the process group of the clip-replica headline; not the headline"}


class _SideGuard:
    """Deadline around a side measurement that runs collectives (frame_sharded, split_clip).  The JSON line is complete before they start; if
    one of them has not returned within `seconds` -- a peer died, the collective sequences diverged, RCCL hangs -- rank 0 prints the line with
    an error in that field and every rank leaves with os._exit(0) (no teardown handshake with peers that may be gone).  One line, once."""

    def __init__(self, line, seconds):
        import threading
        self.line, self.seconds, self._lock, self._done, self._timer, self.key = line, seconds, threading.Lock(), False, None, None

    def arm(self, key):
        import threading
        self.key = key
        self._timer = threading.Timer(self.seconds, self._expired)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def emit(self):
        with self._lock:
            if not self._done and self.line is not None:
                print(json.dumps(self.line), flush=True)
            self._done = True

    def _expired(self):
        if self.line is not None and not self._done:
            self.line[self.key] = {"error": f"no result within {self.seconds} s: a rank failed or a collective did not complete; the fields measured before it stand"}
        self.emit()
        sys.stderr.write(f"bench.py: side measurement `{self.key}` exceeded {self.seconds} s; leaving\n")
        sys.stderr.flush()
        os._exit(0)


def _respawn_ranks(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks as CHILD processes (torch.distributed.run) before
    anything touches the GPU, and exit with their code.  (Never exec: the GPU boxes refuse an exec after GPU init.)"""
    import subprocess
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (default 100 = 5 s of GPU work: long enough for the driver's 5 s GPU-busy sampling)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-splits", action="store_true",
                    help="skip the passes under the other f32-GEMM splits (`alt_f32_splits`): profiling runs that want one policy's kernels only")
    ap.add_argument("--no-in-flight", action="store_true",
                    help="skip the `two_clips_in_flight` side measurement (profiling runs: its overlapping launches would mix into rocprofv3's "
                         "per-kernel averages, which are meant to be compared with the sequential per-launch times of `roofline`)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip `other_configs` (short timed runs of san_online / brivis / brivis_swinl = BASELINE.json configs[2..4] after the headline)")
    ap.add_argument("--inputs", default="device", choices=["device", "host"],
                    help="where the clips live when the timed region starts: device = resident in HBM (the headline), host = pinned host memory, "
                         "uploaded inside every step (the PCIe-inclusive rate; the default run reports it as `host_inputs`)")
    ap.add_argument("--streams", type=int, default=1,
                    help="clips in flight per GPU (openvis_amd.runtime.ClipPipeline: one HIP stream + host thread each); "
                         "the default 1 keeps every launch alone on the GPU, which is what the roofline figures describe -- "
                         "2 raises throughput by ~12 %% at >= 12 steps by filling the small stages of one clip with another's GEMMs")
    ap.add_argument("--clip-precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--model", default="openvis", choices=sorted(MODELS),
                    help="default openvis = the BASELINE.json headline (configs[1]); san_online = configs[2]; brivis = "
                         "configs[3] (one clip of --frames frames, frame-sharded over the ranks, strong scaling)")
    ap.add_argument("--frames", type=int, default=0, help="frames per clip (default 5; brivis: 36)")
    ap.add_argument("--resolution", type=int, default=0, choices=[0, 720, 1080],
                    help="frame height (16:9); default 720, *_swinl models: 1080 (BASELINE.json configs[4])")
    ap.add_argument("--crop-list", default="auto", choices=["auto", "host", "device"],
                    help="MODEL.CLIP_ADAPTER.CROP_LIST: where ClipAdapter's crop list is built (host = the reference's read-back of the boxes in the "
                         "middle of the forward; device = no read-back, static shapes; auto = device while >= 90 %% of the masks are non-empty)")
    ap.add_argument("--gather-masks", action="store_true",
                    help="frame-sharded runs: gather the ten output masks of all frames on rank 0 (the reference's single video_output); "
                         "default: every rank keeps the masks of its own frames for a sharded evaluator (SURVEY.md 8e (3))")
    ap.add_argument("--sharded-frames", type=int, default=36,
                    help="N > 1 with the default model: frames of the ONE BriVIS clip that is additionally run frame-sharded over the ranks "
                         "(`frame_sharded` on the JSON line; BASELINE.json configs[3]: 36); 0 skips it")
    ap.add_argument("--frame-sharded", action="store_true",
                    help="--model san_online / openvis_online with N > 1 (or --process-group): shard ONE clip's frames over the ranks as --model brivis "
                         "does (tracker all-gather, logit all-reduce / crop-logit all-gather) instead of running clip replicas")
    ap.add_argument("--split-frames", type=int, default=8,
                    help="N > 1 (or --process-group) with the default model: frames of the ONE OpenVIS clip that is additionally run split over the "
                         "ranks, split-KV offline decoder (`split_clip` on the JSON line; at least one frame per rank); 0 skips it")
    ap.add_argument("--side-timeout", type=float, default=420.0,
                    help="seconds a multi-rank side measurement (frame_sharded, split_clip) may take before the line is printed without it")
    ap.add_argument("--process-group", action="store_true",
                    help="N = 1: create a ONE-rank RCCL process group anyway and (online models) run the frame-sharded control flow over it -- "
                         "the all-gather on the side stream, the logit all-reduce, the mask gather -- so that the RCCL code path executes on a "
                         "one-GPU box (tests/test_sharded_gpu.py)")
    ap.add_argument("--precision", default="mixed", choices=["mixed", "fp32"],
                    help="dense-path policy: mixed = the reference's autocast policy, fp32 = exact f32 everywhere")
    ap.add_argument("--f32-split", default="auto", choices=["auto", "fp16x2", "bf16x3", "bf16x2", "f32"],
                    help="MODEL.F32_GEMM_SPLIT: how large f32 GEMMs reach the 16-bit MFMA (auto: the f32-grade fp16x2 under --precision "
                         "mixed, the f32-grade bf16x3 under fp32; bf16x2 = 16 significand bits per operand, an explicit opt-in)")
    args = ap.parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != env_world:
        if env_world == 1 and args.gpus > 1:
            _respawn_ranks(args)                              # does not return
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    from openvis_amd import distributed as D
    # OVIS_BENCH_TEST_RIG=1 (tests only): all ranks on cuda:0 with a gloo rendezvous, to exercise the N > 1 control flow on a
    # one-GPU box; the driver's multi-GPU runs use one GPU per rank over RCCL
    rig = os.environ.get("OVIS_BENCH_TEST_RIG") == "1"
    torch.cuda.set_device(0 if rig else int(os.environ.get("LOCAL_RANK", "0")))
    if env_world > 1 or args.process_group:
        # RCCL prints a version banner ("RCCL version : ...", five lines) to the C stdout when its first communicator comes up.  stdout of this
        # command is ONE JSON line: file descriptor 1 is pointed at stderr for every native library, python's print keeps the real stdout
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real_stdout, "w", buffering=1)
    if env_world > 1 and "OMP_NUM_THREADS" not in os.environ:
        # one process per GPU on one host: a default-sized intra-op pool per rank (one thread per core, spinning after every parallel region)
        # oversubscribes the host N times over -- measured on the test rig: model set-up 40 s instead of 5 s with two ranks
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // env_world)))
    rank, world, local_rank = D.init_from_env("gloo" if rig else "nccl", force=args.process_group)   # "nccl" is RCCL on ROCm
    device = torch.device("cuda", 0 if rig else local_rank)

    from openvis_amd import ops
    f32_split = args.f32_split if args.f32_split != "auto" else ("bf16x3" if args.precision == "fp32" else "fp16x2")
    model, sd, text = build_model(device, clip_precision=args.clip_precision, precision=args.precision, model_name=args.model, f32_split=args.f32_split,
                                  crop_list=args.crop_list)
    # one clip over the ranks: BriVIS always (configs[3] is a 36-frame clip); the other two per-frame architectures on request
    frame_sharded = ((args.model.startswith("brivis") or (args.frame_sharded and args.model in ("san_online", "openvis_online")))
                     and (world > 1 or args.process_group))
    T = args.frames or (36 if args.model.startswith("brivis") else T_CLIP)
    if frame_sharded and T < world:
        raise SystemExit(f"bench.py: a {T}-frame clip cannot be frame-sharded over {world} ranks (a rank would own no frame): pass --frames >= {world}")
    res = args.resolution or (1080 if args.model.endswith("_swinl") else 720)
    FH, FW = res, res * 16 // 9                          # frame size of this run
    fwd_kw = {}
    # `value`: the clips are RESIDENT IN HBM when the timed region starts (uint8 [T,3,H,W] device tensors; the model takes them as views).
    # `--inputs host`, and the `host_inputs` side measurement of the default run: the clips live in PINNED HOST memory (what a
    # DataLoader(pin_memory=True) hands to the model) and every step moves its uint8 frames (13.8 MB per 720p clip) over PCIe inside
    # model.forward -- the PCIe-inclusive rate, reported beside the headline, never as it.
    def place(c):
        return c.to(device) if args.inputs == "device" else c.pin_memory()
    if frame_sharded:
        # ONE clip, contiguous frame blocks per rank, all-gather of query embeddings before the linker
        fr = D.inference_shard(T, rank, world)
        fwd_kw = {"frame_range": (fr.start, fr.stop)}
        if args.gather_masks:
            fwd_kw["gather_masks_to"] = 0
        host_clips = [synth_frames(T, FH, FW, 1000 + i, "cpu") for i in range(2)]
    else:
        # clip-level sharding (InferenceSampler layout): 2*world clips, each rank owns a contiguous shard
        my_clips = D.inference_shard(2 * world, rank, world)
        host_clips = [synth_frames(T, FH, FW, 1000 + i, "cpu") for i in my_clips]
    clips = [place(c) for c in host_clips]
    inputs = [[{"image": [f for f in c], "dataset_name": "synthetic_burst_val"}] for c in clips]
    _model = model
    model = (lambda inp, **kw: _model(inp, **fwd_kw, **kw)) if fwd_kw else _model

    def sync_all():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    # frames every rank processes per step (frame-sharded: its block of the clip; clip replicas: whole clips)
    frames_per_rank = ([len(D.inference_shard(T, r, world)) for r in range(world)] if frame_sharded else [T] * world)
    if world > 1 or args.process_group:
        # RCCL builds its rings / channels at the first collective of every kind, so the first ones run OUTSIDE the timed region:
        # clip replicas only ever use the scalar all-reduce of the timing; the frame-sharded path also its all-gather / all-reduce /
        # gather (distributed.warm_up; the warm-up steps then repeat them through the model)
        D.max_over_ranks(0.0, "cpu" if rig else device)
        if frame_sharded:
            D.warm_up(device if not rig else "cpu")

    if f32_split == "fp16x2":
        # the timed loop drops its outputs unread (only the last one is waited for), so the range flags of ALL forwards are OR-ed on the device
        from openvis_amd.modeling.video_maskformer import StickyFlag
        _model.sticky_range_flag = StickyFlag(device)          # one device word per host thread (clips in flight), OR-ed at readout
    out = None
    if args.streams > 1 and not frame_sharded:
        # K steps = K clips, `--streams` of them in flight (independent clips; results identical to the sequential loop)
        from openvis_amd.runtime import ClipPipeline
        pipe = ClipPipeline(_model, args.streams)
        pipe.run([inputs[i % len(inputs)] for i in range(max(args.warmup, 1))])
        sync_all()
        t0 = time.perf_counter()
        out = pipe.run([inputs[i % len(inputs)] for i in range(args.steps)])[-1]
        sync_all()
        elapsed = time.perf_counter() - t0
    else:
        for i in range(args.warmup):
            out = model(inputs[i % len(inputs)])
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = model(inputs[i % len(inputs)])
        sync_all()
        elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, "cpu" if rig else device)
    if out is not None and hasattr(out, "wait"):
        out.wait()                                             # reads the last clip's outputs (and, under fp16x2, its range flag)
    # an activation left the fp16 range in ANY timed / warm-up forward (sticky flag), or the model already went to bf16x3 on one that was read
    fell_back = f32_split == "fp16x2" and (_model.f32_gemm_mode != 3 or int(_model.sticky_range_flag.item()) != 0)

    # ---- the same step with the OTHER f32-GEMM splits, timed the same way (shorter) --------------------------------------------
    # The pixel decoder / masked-attention decoder are f32 in the reference (msdeformattn.py:329 disables autocast).  Their large
    # GEMMs run on the 16-bit MFMA as the f32-grade fp16x2 (3 fp16 products of the 11 + 11 bit split; the default under --precision
    # mixed), as the f32-grade bf16x3 (6 bf16 products) or as bf16x2 (3 bf16 products, 16 significand bits per operand: NOT f32-grade,
    # listed for comparison only).  Whichever the headline uses, the others stand beside it in `alt_f32_splits`.
    alts = []
    if not (args.streams > 1 and not frame_sharded) and f32_split != "f32" and not args.no_alt_splits:
        from openvis_amd.config import F32_GEMM_SPLITS
        keep_mode = _model.f32_gemm_mode
        for alt_name in [n for n in ("fp16x2", "bf16x3", "bf16x2") if n != f32_split]:
            _model.f32_gemm_mode = F32_GEMM_SPLITS[alt_name]
            n_alt = max(args.steps // 2, 1)
            for i in range(2):
                model(inputs[i % len(inputs)])
            sync_all()
            t0 = time.perf_counter()
            for i in range(n_alt):
                model(inputs[i % len(inputs)])
            sync_all()
            e_alt = D.max_over_ranks(time.perf_counter() - t0, "cpu" if rig else device)
            alts.append({"split": alt_name, "f32_grade": alt_name != "bf16x2",
                         "value": round(T * n_alt * (1 if frame_sharded else world) / e_alt, 3), "unit": "frames/s",
                         "ms_per_step": round(e_alt / n_alt * 1e3, 3), "steps": n_alt})
        _model.f32_gemm_mode = keep_mode
        out = model(inputs[0])                                 # back on the headline's split before the profiling passes
        torch.cuda.synchronize()
    alt = alts[0] if alts else None

    # ---- the same clips with an f32-class BACKBONE (the parity-grade policy), timed the same way ---------------------------------
    # The headline's ResNet runs fp16 MFMA operands with f32 accumulation (the reference's autocast, train_net.py:241), which moves mask
    # logits by up to 2e-2: on C2 its masks differ from the f32 oracle in ~14 k of 29.4 M bits, all inside |logit| < 3e-2.  With the
    # backbone under MODEL.BACKBONE_PRECISION = fp32 (large convs on the f32-grade split like the pixel decoder) the same clip differs in
    # ~120 bits, <= 5 of them beyond |logit| 1e-3 (tests/test_c2_720p_gpu.py, profiles/r06/parity_summary.txt).  That policy's frames/s
    # stands here, driver-timed, next to the headline.
    alt_bb = None
    if (args.model == "openvis" and _model.backbone.precision == "fp16" and not (args.streams > 1) and not args.no_alt_splits
            and hasattr(_model.backbone, "h16_storage")):
        from openvis_amd.modeling.backbone.resnet import ResNet
        keep_bb = _model.backbone
        _model.backbone = ResNet(keep_bb.depth, keep_bb.out_features, precision="fp32").load_state_dict(sd, "backbone.", device)
        n_alt = max(args.steps // 2, 1)
        for i in range(2):
            model(inputs[i % len(inputs)])
        sync_all()
        t0 = time.perf_counter()
        for i in range(n_alt):
            model(inputs[i % len(inputs)])
        sync_all()
        e_alt = D.max_over_ranks(time.perf_counter() - t0, "cpu" if rig else device)
        alt_bb = {"backbone": f"f32-class ({f32_split} on the 16-bit MFMA, native f32 for the small layers)", "value": round(T * n_alt * world / e_alt, 3),
                  "unit": "frames/s", "ms_per_step": round(e_alt / n_alt * 1e3, 3), "steps": n_alt,
                  "parity": "C2 full size: <= 16 mask bits beyond |oracle logit| 1e-3 (asserted; measured 5), 0 beyond 3e-2; headline policy: 0 beyond 3e-2"}
        _model.backbone = keep_bb
        del keep_bb
        out = model(inputs[0])
        torch.cuda.synchronize()

    # ---- the same steps with the clips in PINNED HOST memory: the PCIe-inclusive rate ----------------------------------------------------
    host_inputs = None
    if args.inputs == "device" and not (args.streams > 1) and not frame_sharded and not args.no_alt_splits:
        pinned = [[{"image": [f for f in c.pin_memory()], "dataset_name": "synthetic_burst_val"}] for c in host_clips]
        n_h = max(args.steps // 2, 1)
        for i in range(2):
            model(pinned[i % len(pinned)])
        sync_all()
        t0 = time.perf_counter()
        for i in range(n_h):
            model(pinned[i % len(pinned)])
        sync_all()
        e_h = D.max_over_ranks(time.perf_counter() - t0, "cpu" if rig else device)
        host_inputs = {"value": round(T * n_h * world / e_h, 3), "unit": "frames/s", "ms_per_step": round(e_h / n_h * 1e3, 3), "steps": n_h,
                       "note": f"clips in pinned host memory, {host_clips[0].numel() / 1e6:.1f} MB of uint8 frames cross PCIe inside every step; not the headline"}
        del pinned
        out = model(inputs[0])
        torch.cuda.synchronize()

    # ---- the same clips with TWO in flight (openvis_amd.runtime.ClipPipeline: one HIP stream + host thread per slot) --------------------
    # What a serving process would run: the tails of the persistent GEMMs and the latency-bound decoder of one clip fill with the other
    # clip's kernels.  Reported BESIDE the headline, not as it: per-launch HIP-event times of overlapping clips include each other's kernels,
    # so the `roofline` below could not be read from such a timed region (`--streams 2` makes it the timed region).
    in_flight2 = None
    if args.streams == 1 and not frame_sharded and not rig and not args.no_in_flight:
        from openvis_amd.runtime import ClipPipeline
        pipe2 = ClipPipeline(_model, 2)
        n2 = max(args.steps // 2, 4)
        pipe2.run([inputs[i % len(inputs)] for i in range(3)])
        sync_all()
        t0 = time.perf_counter()
        pipe2.run([inputs[i % len(inputs)] for i in range(n2)])
        sync_all()
        e2 = D.max_over_ranks(time.perf_counter() - t0, device)
        in_flight2 = {"value": round(T * n2 * world / e2, 3), "unit": "frames/s", "ms_per_step": round(e2 / n2 * 1e3, 3), "steps": n2,
                      "note": "2 clips in flight per GPU (HIP streams), same clips and outputs; not the headline"}
        torch.cuda.synchronize()

    # ---- frame-sharded runs: what every rank spends in the exchange steps, untimed extra steps --------------------
    collective_ms = collective_spans(model, inputs) if frame_sharded else None

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream -----------
    # Per-launch events around every GEMM / K1 launch (ops.PROFILE).  Two untimed passes:
    #   in situ : the K clips again exactly as in the timed region (with `--streams` > 1 the kernels of different clips
    #             share the GPU, so a launch lasts longer than alone -- this is what rocprofv3 sees for this command);
    #   isolated: one clip alone on one stream (the kernel's own rate; also fills the per-stage tensors `st`).
    pipelined = args.streams > 1 and not frame_sharded
    agg_situ = hbm_situ = None
    if pipelined:
        ops.PROFILE = []
        pipe.run([inputs[i % len(inputs)] for i in range(args.steps)])
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        agg_situ, hbm_situ = _aggregate(prof)
    ops.PROFILE = []
    st = {}
    model(inputs[0], stages=st)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg, hbm = _aggregate(prof)                              # isolated
    n_situ = args.steps if pipelined else 1                  # clips behind the in-situ sums
    if not pipelined:
        agg_situ, hbm_situ = agg, hbm
    # "The dominant kernel" = the kernel TEMPLATE with the largest summed launch time (rocprofv3 lists every instantiation as
    # its own row -- <out dtype, activation, residual> for the fp16 GEMM --, but it is one piece of code); the line carries
    # the aggregate over its instantiations and every instantiation's own row for the cross-check against rocprofv3.
    fam_situ, fam_iso = _families(agg_situ), _families(agg)
    kbase, (n_launch, flops, secs, members) = max(fam_situ.items(), key=lambda kv: kv[1][2])
    dom = (_family_label(kbase, members), None)
    achieved = flops / secs / 1e12
    iso = fam_iso.get(kbase, (n_launch, flops, secs))
    peak, peak_note = _family_peak(kbase, f32_split)
    # HBM traffic of that kernel: PMC passes of this same command (tools/pmc_traffic.py: FETCH_SIZE x2 gfx950 correction +
    # WRITE_SIZE, separate --pmc runs) committed under profiles/rNN/ -- a STATIC figure (bench.py cannot run rocprofv3 on
    # itself), tagged with the file and the commit it was collected at; null if no summary has this kernel
    def _norm(k):
        return k.replace("(anonymous namespace)::", "").replace("void ", "").replace(" ", "")

    pmc, pmc_src, pmc_stale = {}, None, None
    import glob
    digest = csrc_digest()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic_bench.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("csrc_sha256") != digest:
            # counters of OTHER kernel sources say nothing about the kernels timed here: traffic stays null, with the reason on the line
            pmc_stale = {"file": os.path.relpath(f, ROOT), "commit": d.get("commit"),
                         "reason": "openvis_amd/csrc changed since these PMC passes were collected (csrc_sha256 differs): re-run tools/final_verify.sh"}
            break
        pmc = {_norm(k): v for k, v in d["kernels"].items()}
        pmc_src = {"file": os.path.relpath(f, ROOT), "commit": d.get("commit"), "csrc_sha256": digest[:16], "static": True,
                   "note": "PMC passes of this same command on the same kernel sources (digest verified at run time)"}
        break

    def _pmc_lookup(name):
        fh = name.endswith(",FH>")                            # fp16x2 instantiations: the template's LAST argument is true
        name = name.replace(",FH>", ">")
        v = pmc.get(_norm(name)) if not fh else None
        if v is None and "<" not in name:                     # un-templated kernels: prefix match
            v = next((x for k, x in pmc.items() if k.startswith(_norm(name))), None)
        if v is None and "<" in name:
            # the bench label names <BM,BN,LoaderA>; rocprofv3 also prints the B loader / plane count (and the ovis:: scope):
            # launch-weighted mean over the instantiations that share the label's template arguments
            def _targs(k):
                k = k.replace("ovis::", "")
                if "<" not in k:
                    return k, []
                b, rest = k.split("<", 1)
                out, depth, cur = [], 0, ""
                for ch in rest[:rest.rfind(">")]:
                    if ch == "," and depth == 0:
                        out.append(cur)
                        cur = ""
                        continue
                    depth += ch == "<"
                    depth -= ch == ">"
                    cur += ch
                return b, out + [cur]
            base, targs = _targs(_norm(name))
            hits = []
            for k, x in pmc.items():
                b, ta = _targs(k)
                if not (b == base and len(ta) >= len(targs) and all(a.startswith(t) for a, t in zip(ta, targs))):
                    continue
                is_fh = ((b == "gemm_f16_pp_kernel" and len(ta) >= 13 and ta[12] == "true") or           # FH (fp16x2): template argument 13
                         (b == "gemm_f32x3_kernel" and len(ta) >= 6 and ta[5] == "true"))                 # ... and 6 of these two kernels
                if is_fh != fh:
                    continue
                hits.append(x)
            if hits:
                nl = sum(h["launches"] for h in hits)
                v = {f: round(sum(h[f] * h["launches"] for h in hits) / nl) for f in
                     ("hbm_bytes_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch")}
                v["launches"] = nl
        return v

    # per launch like `achieved`: launch-weighted mean over the instantiations
    tvs = [(_pmc_lookup(k), v[0]) for k, v in members.items()]
    tv = None
    if tvs and all(t is not None for t, _ in tvs):
        nl = sum(n for _, n in tvs)
        tv = {f: round(sum(t[f] * n for t, n in tvs) / nl) for f in ("hbm_bytes_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch")}
    traffic = tv["hbm_bytes_per_launch"] if tv else None
    roofline = {"kernel": dom[0], "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "peak_note": peak_note, "traffic": traffic,
                "traffic_source": pmc_src if traffic is not None else pmc_stale,
                "traffic_split": ({"fetch": tv["fetch_bytes_per_launch"], "write": tv["write_bytes_per_launch"]} if tv else None),
                "measured": (f"HIP events per launch, in situ: {args.steps} clips with {args.streams} in flight (launches of "
                             "different clips share the GPU)" if pipelined else "HIP events per launch, one clip on one stream"),
                "launches_per_step": n_launch // n_situ, "avg_launch_ms": round(secs / n_launch * 1e3, 4),
                "gflop_per_launch": round(flops / n_launch / 1e9, 2),
                "isolated": {"achieved": round(iso[1] / iso[2] / 1e12, 2), "frac": round(iso[1] / iso[2] / 1e12 / peak, 4),
                             "avg_launch_ms": round(iso[2] / iso[0] * 1e3, 4),
                             "note": "the same kernel with one clip alone on the GPU"},
                "instantiations": {k: {"launches_per_step": v[0] // n_situ, "avg_launch_ms": round(v[2] / v[0] * 1e3, 4),
                                       "gflop_per_launch": round(v[1] / v[0] / 1e9, 2), "achieved": round(v[1] / v[2] / 1e12, 1),
                                       "frac": round(v[1] / v[2] / 1e12 / peak, 4),
                                       "traffic": (_pmc_lookup(k) or {}).get("hbm_bytes_per_launch"),
                                       # PMC bytes / this launch time: how close the instantiation sits to the HBM roof instead
                                       "traffic_gbps": (round((_pmc_lookup(k) or {}).get("hbm_bytes_per_launch") / (v[2] / v[0]) / 1e9, 1)
                                                        if (_pmc_lookup(k) or {}).get("hbm_bytes_per_launch") else None)}
                                   for k, v in sorted(members.items(), key=lambda kv: -kv[1][2])},
                "all_gemm_kernels": {k: {"launches": v[0], "ms": round(v[2] * 1e3, 3), "TFLOPs": round(v[1] / v[2] / 1e12, 1)}
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])},
                "all_gemm_kernels_note": "one isolated clip: launches, summed ms, rate"}

    # stage breakdown (untimed extra pass, OpenVIS only): wall time of each stage with a device sync after it
    stage_ms = None
    if args.model == "openvis" and world == 1:
        def _t(fn, reps=2):                                   # best of 2: the pass is outside the timed region
            best, r = None, None
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) * 1e3
                best = dt if best is None else min(best, dt)
            return r, best
        m = _model
        inp = inputs[0]
        names = m.get_class_name_list(inp[0]["dataset_name"])
        frames, t_in = _t(lambda: m._frames_to_device(inp))
        (images, image_size, padded), t_a1 = _t(lambda: m.preprocess(frames))
        feats, t_a2 = _t(lambda: m.backbone(images))
        (mf_out), t_a36 = _t(lambda: m.sem_seg_head.pixel_decoder.forward_features(feats))
        outs, t_a78 = _t(lambda: m.sem_seg_head.predictor(mf_out[2], mf_out[0]))
        (probs, row_ids, _e), t_clip = _t(lambda: m.open_vocabulary_inference(outs["pred_logits"][0], outs["pred_masks"][0], frames, names, padded))
        _, t_out = _t(lambda: m.inference_video(m.num_queries, len(names), probs, row_ids, outs["pred_masks"][0], padded, image_size,
                                                 image_size[0], image_size[1]))
        stage_ms = {"A1_preprocess": round(t_a1, 2), "A2_backbone": round(t_a2, 2), "A3-A6_pixel_decoder": round(t_a36, 2),
                    "A7-A8_decoder": round(t_a78, 2), "A9-A12_boxes_crops_clip_logits": round(t_clip, 2),
                    "A16_topk_masks_d2h": round(t_out, 2)}

    bb_name = {"r50": "R50", "swin_l": "Swin-L"}[MODELS[args.model].get("backbone", "r50")]
    bb_prec = _model.backbone.precision
    # K1 (deformable sampling): HBM roofline of the gather kernel, same live HIP-event measurement
    roofline_k1 = None
    if hbm_situ:
        kname, (kn, kbytes, ksecs) = max(hbm_situ.items(), key=lambda kv: kv[1][2])
        gbs = kbytes / ksecs / 1e9
        kiso = hbm.get(kname, (kn, kbytes, ksecs))
        kv = _pmc_lookup(kname)
        ktraffic = kv["hbm_bytes_per_launch"] if kv else None
        roofline_k1 = {"kernel": kname, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": ktraffic, "launches_per_step": kn // n_situ,
                       "avg_launch_ms": round(ksecs / kn * 1e3, 4), "algorithmic_mb_per_launch": round(kbytes / kn / 1e6, 2),
                       "isolated": {"achieved": round(kiso[1] / kiso[2] / 1e9, 1), "avg_launch_ms": round(kiso[2] / kiso[0] * 1e3, 4)}}
        # The roof that BINDS K1 is not HBM (the value tensor of a frame level is L2 / Infinity-Cache resident): every (query, head, level, point)
        # pulls four 128-byte tap segments through the CU's L1 -- 48 taps x 128 B x 8 heads per token.  MI355X_MICROARCH.md, "Indexed rows: gather
        # into LDS": rows served from the XCD's L2 arrive at 66-73 GB/s per CU = 16.8-18.8 TB/s chip-wide (profiles/r05/k1_tiled_sweep.txt).
        # constants from the model's own deformable attention (d_model C, heads M, levels L, points P), not hard-wired: algorithmic bytes per
        # token = 4 (2 C + 3 M L P) (f32 value row + output row + offsets / logits: 3 200 B for 256 / 8 / 3 / 4); a (query,
        # head, level, point) gathers four taps of C / M f32 channels
        sa = _model.sem_seg_head.pixel_decoder.layers[0].self_attn
        Cm, Mh, Ll, Pp = sa.d_model, sa.n_heads, sa.n_levels, sa.n_points
        tokens = kbytes / kn / (4.0 * (2 * Cm + 3 * Mh * Ll * Pp))
        gathered = tokens * Mh * (Ll * Pp * 4) * (Cm // Mh * 4)
        roofline_k1["gather"] = {"bound": "L2 -> CU gather path", "gathered_bytes_per_launch": int(gathered),
                                 "achieved": round(gathered / (ksecs / kn) / 1e12, 2), "peak": 18.8, "unit": "TB/s",
                                 "frac": round(gathered / (ksecs / kn) / 1e12 / 18.8, 3),
                                 "peak_note": "16.8-18.8 TB/s: the guide's measured rate for rows gathered from the XCD's L2 (66-73 GB/s per CU)"}
    line = None
    if rank == 0:
        frames_total = T * args.steps * (1 if frame_sharded else world)
        n_valid = int(st["valid"].sum()) if "valid" in st else 0
        line = {
            "metric": ("frames/sec (whole node) OpenVIS R50 720p inference" if args.model == "openvis" and res == 720 else
                       f"frames/sec (whole node) {args.model} {bb_name} {res}p inference"), "value": round(frames_total / elapsed, 3),
            "unit": "frames/s", "n_gpus": world, "world_size_seen": D.world_size(), "frames_per_rank": frames_per_rank,
            "process_group": D.backend_name(),
            "steps": args.steps, "warmup": args.warmup, "f32_split": f32_split, "f32_split_f32_grade": f32_split != "bf16x2",
            "f32_split_bound": {"fp16x2": "error <= max(2^-22 |a|, 2^-29) per activation (absolute floor of the fp16 lo plane at a_scale 16) and 2^-22 |w| per weight: "
                                          "f32-grade against the row scale |A||W| + |b| + |R| (2.0-3.0e-7 measured, native f32 MFMA 3.6-4.5e-7), not a relative bound per element",
                                "bf16x3": "exact 3-way bf16 split, six products: relative ~1e-7", "bf16x2": "16 significand bits per operand: relative < 2^-16",
                                "f32": "native f32 MFMA"}.get(f32_split),
            "f32_split_fell_back_to_bf16x3": bool(fell_back),
            "crop_list": getattr(getattr(_model, "clip_adapter", None), "crop_list", None), "alt_f32_split": alt, "alt_f32_splits": alts,
            "inputs": ("resident in HBM (uint8 [T,3,H,W] device tensors) when the timed region starts" if args.inputs == "device" else
                       "pinned host memory: every step uploads its frames (PCIe-inclusive)"), "host_inputs": host_inputs,
            "alt_backbone_f32": alt_bb, "two_clips_in_flight": in_flight2, "frame_sharded": None, "split_clip": None,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if frame_sharded else "weak",
            "vs_baseline": None, "dtype": "f16" if "fp16" in (bb_prec, _model.clip_adapter.precision) else "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} {bb_name} {res}p ({FH}x{FW} -> {(FH + 31) // 32 * 32}x{(FW + 31) // 32 * 32}), "
                                   f"{MODELS[args.model].get('queries', 100)} queries, 482 classes, {T}-frame clips, "
                                   + ("ClipAdapter" if args.model.startswith("openvis") else "SideAdapter")
                                   + f" {MODELS[args.model].get('clip', 'ViT-B/16')}, random-init weights", "frames_per_step": T,
                       "precision": (f"backbone GEMM operands {bb_prec}" + (" / f32 accumulate (the reference's autocast)" if bb_prec == "fp16" else
                                     " (exact 3-way bf16 split on the bf16 MFMA)") + "; pixel decoder, masked-attention decoder, masks and "
                                     "logits f32; CLIP ViT GEMM operands " +
                                     ("fp16 with f32 accumulation (the reference's GPU CLIP dtype)"
                                      + (", ln_1 / ln_2 folded into in_proj / c_fc with f32 statistics" if getattr(getattr(_model.clip_adapter, "visual", None), "fold_ln", False)
                                         and getattr(_model.clip_adapter.visual, "stream16", False) else "")
                                      if _model.clip_adapter.precision == "fp16" else "f32") + (f"; resampler {_model.resampler.precision}" if hasattr(_model, "resampler") else "")
                                     + f"; large f32 GEMMs/convs as {f32_split} on the 16-bit MFMA (MODEL.F32_GEMM_SPLIT)"),
                       "valid_crops_per_clip": n_valid,
                       "parallelism": (f"frame-sharded x{world} + RCCL all-gather" if frame_sharded else f"clip-replicas x{world}")
                                      + (f", {args.streams} clips in flight per GPU (HIP streams)" if args.streams > 1 and not frame_sharded else "")},
            "roofline": roofline, "roofline_k1": roofline_k1, "stage_ms": stage_ms, "collective_ms": collective_ms,
            "stage_ms_note": "one clip alone, device sync after every stage (clips in flight overlap these stages)",
        }
        if (world == 1 and args.model == "openvis" and res == 720 and not args.no_other_configs and args.streams == 1
                and D.backend_name() is None):
            # BASELINE.json configs[2], [3], [4] on the driver's line: short runs after the headline, each with its own dominant-kernel roofline
            del out, st
            _model = model = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            line["other_configs"] = [measure_other_config(n, device, args, f32_split) for n in ("san_online", "brivis", "brivis_swinl")]
            line["other_configs_note"] = ("N = 1 runs of the other BASELINE.json configs, same timing rules as the headline (warm-up, then `steps` "
                                          "steps between device synchronisations, clips resident in HBM, D2H of the masks included); not the headline")
        if not args.no_cpu_baseline and world == 1 and args.model == "openvis":
            line["cpu_baseline"] = cpu_baseline(sd)

    # ---- N > 1 (or --process-group), default model: the two single-clip splits next to the clip-replica headline, LAST and guarded ----------
    # The OpenVIS offline decoder attends over all frames of a clip, so the headline shards by CLIP and its only collective is the timing's
    # scalar.  What moves data over RCCL / xGMI is (a) the frame-sharded BriVIS pass (configs[3]) and (b) ONE OpenVIS clip over the ranks
    # (split-KV decoder): one command measures all three.  Both run after every headline field exists, under a deadline (_SideGuard): a rank
    # that fails or a collective that hangs costs the side field, never the line.
    sides = []
    if world > 1 and not frame_sharded and args.model == "openvis" and args.sharded_frames > 0:
        if world > args.sharded_frames:
            # more ranks than frames: inference_shard() would hand some ranks an EMPTY frame block and they would fall out of the collective
            # sequence (frames[0] of an empty list) -- skipped on every rank alike, with the reason on the line
            if rank == 0:
                line["frame_sharded"] = {"skipped": f"--sharded-frames {args.sharded_frames} < world size {world}: a rank would own no frame"}
        else:
            sides.append(("frame_sharded", lambda: measure_frame_sharded(device, rank, world, rig, args, sync_all)))
    if (world > 1 or args.process_group) and not frame_sharded and args.model == "openvis" and args.split_frames > 0 and args.streams == 1:
        sides.append(("split_clip", lambda: measure_split_clip(_model, device, rank, world, rig, args, sync_all)))
    guard = _SideGuard(line if rank == 0 else None, args.side_timeout)
    diverged = False
    for key, fn in sides:
        guard.arm(key)
        try:
            res = fn()
        except Exception as e:                               # this rank left the collective sequence: no further collectives from here on
            res, diverged = {"error": f"{type(e).__name__}: {e}"[:400]}, True
        guard.disarm()
        if rank == 0:
            line[key] = res
        if diverged:
            break
    if rank == 0:
        guard.emit()
    if diverged:
        sys.stdout.flush()
        os._exit(0)                                          # the peers are (or will be) stuck in a collective this rank never joins: no teardown handshake
    if D.backend_name() is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
