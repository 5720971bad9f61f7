"""Multi-GPU plumbing for the clip-sharded eval path (one process per GPU, torch.distributed; "nccl" == RCCL on ROCm).

OpenVIS' offline decoder attends over every frame of a clip, so the unit of parallelism is the CLIP: ranks own
contiguous shards of the clip list exactly like detectron2's InferenceSampler (openvis/data/build.py:238-247) and
there is no collective on the data path.  The only collective is the scalar MAX used for timing.

The online models (BriVIS) also shard ONE clip by frames: all_gather_frames_async (query embeddings, on a side stream),
all_reduce_sum (per-query logit sums) and gather_frame_masks (the selected output masks to the output rank).  Ranks are
started by torch.distributed.run / bench.py's child spawn BEFORE anything touches the GPU, and every rank calls
torch.cuda.set_device(LOCAL_RANK) before its first HIP call (bench.py main)."""
import os

import torch


def init_from_env(backend=None, force=False):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run). Returns
    (rank, world, local_rank); a no-op for world == 1 unless `force` (or OVIS_FORCE_PROCESS_GROUP=1): a ONE-rank group is legal, and on
    a one-GPU box a 1-rank "nccl" group is the only way to execute the RCCL branches of this module (async all_gather_into_tensor on the
    side stream, device all-reduce, dist.gather of device tensors) before the first multi-GPU run does."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("OVIS_FORCE_PROCESS_GROUP") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, world, local_rank


def inference_shard(total, rank, world):
    """Contiguous shard of range(total) owned by `rank` (detectron2 InferenceSampler._get_local_indices)."""
    shard = total // world
    left = total % world
    sizes = [shard + int(r < left) for r in range(world)]
    begin = sum(sizes[:rank])
    return range(begin, min(begin + sizes[rank], total))


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (the bench's elapsed time)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def world_size():
    """World size the process group actually has (1 without one) -- bench.py prints it next to --gpus."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def backend_name():
    """"nccl" (= RCCL) / "gloo" of the default process group, None without one."""
    import torch.distributed as dist
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


# ---- per-rank timing of the exchange steps (bench.py --model brivis --gpus N prints them as `collective_ms`) ---------------------------
SPANS = None          # None: off.  dict name -> list of (start, end): HIP events on the calling stream (device work) or perf_counter pairs


class span:
    """`with span("all_gather_wait"): ...` -- records how long the enclosed work takes ON THE CURRENT STREAM (HIP events; the host does not
    wait), or host wall time for host-staged collectives (gloo rigs).  Off (no events, no cost) unless SPANS is a dict."""

    def __init__(self, name, host=False):
        self.name, self.host = name, host

    def __enter__(self):
        if SPANS is not None:
            if self.host or not torch.cuda.is_available():
                import time
                self.t0 = time.perf_counter()
            else:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.e0.record()
        return self

    def __exit__(self, *a):
        if SPANS is not None:
            if hasattr(self, "t0"):
                import time
                SPANS.setdefault(self.name, []).append((self.t0, time.perf_counter()))
            else:
                self.e1.record()
                SPANS.setdefault(self.name, []).append((self.e0, self.e1))


def spans_ms():
    """mean milliseconds per recorded span name (call after torch.cuda.synchronize())."""
    out = {}
    for k, v in (SPANS or {}).items():
        ms = [(b - a) * 1e3 if isinstance(a, float) else a.elapsed_time(b) for a, b in v]
        out[k] = round(sum(ms) / max(len(ms), 1), 4)
    return out


def gather_objects(obj):
    """every rank's python object on every rank (bench diagnostics; a list of one without a process group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra HIP stream per device for collectives that overlap the compute stream."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class _FramesGather:
    """Handle of an all-gather of per-frame rows that may still be in flight; wait() -> [total_frames, ...] on the
    caller's current stream."""

    def __init__(self, out, sizes, tmax, device, work=None, side=None):
        self.out, self.sizes, self.tmax, self.device, self.work, self.side = out, sizes, tmax, device, work, side

    def wait(self):
        if self.work is not None:
            self.work.wait()                                    # the CURRENT stream waits for the collective
            torch.cuda.current_stream().wait_stream(self.side)
            self.out.record_stream(torch.cuda.current_stream())
            self.work = None
        if self.sizes is None:
            return self.out
        o = self.out.view((len(self.sizes), self.tmax) + tuple(self.out.shape[1:]))
        if all(n == self.tmax for n in self.sizes):
            return o.reshape((-1,) + tuple(self.out.shape[1:])).to(self.device)
        return torch.cat([o[r, :n] for r, n in enumerate(self.sizes)], dim=0).to(self.device)


def all_gather_frames_async(local, total_frames):
    """All-gather of per-frame rows over contiguous frame shards (the ONE exchange of the frame-sharded
    path): `local` is this rank's [t_local, ...] block of a [total_frames, ...] tensor sharded with `inference_shard`.
    Blocks are padded to the largest shard so ONE all_gather_into_tensor moves everything (512 KB per rank for [5,100,256]
    f32: latency-bound, ~4 MB over xGMI).  On RCCL the pad + collective run on a SIDE stream, so whatever the caller
    launches on its own stream before wait() (the CLIP back pass of the local frames) overlaps the exchange."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return _FramesGather(local, None, 0, local.device)
    world = dist.get_world_size()
    sizes = [len(inference_shard(total_frames, r, world)) for r in range(world)]
    tmax = max(sizes)
    assert local.shape[0] == sizes[dist.get_rank()], (local.shape, sizes, dist.get_rank())
    dev = local.device
    if dist.get_backend() == "gloo":                            # CPU tests / one-GPU test rigs: synchronous, staged through the host
        pad = torch.zeros((tmax,) + tuple(local.shape[1:]), dtype=local.dtype)
        pad[: local.shape[0]] = local.detach().cpu()
        out = torch.empty((world * tmax,) + tuple(local.shape[1:]), dtype=local.dtype)
        dist.all_gather_into_tensor(out, pad)
        return _FramesGather(out, sizes, tmax, dev)
    side, main = _side_stream(dev), torch.cuda.current_stream()
    side.wait_stream(main)                                      # `local` is produced on the caller's stream
    with torch.cuda.stream(side):
        pad = local.new_zeros((tmax,) + tuple(local.shape[1:]))
        pad[: local.shape[0]] = local
        out = local.new_empty((world * tmax,) + tuple(local.shape[1:]))
        work = dist.all_gather_into_tensor(out, pad, async_op=True)
    local.record_stream(side)
    return _FramesGather(out, sizes, tmax, dev, work, side)


def all_gather_frames(local, total_frames):
    """Synchronous form of all_gather_frames_async."""
    return all_gather_frames_async(local, total_frames).wait()


def gather_frame_masks(masks, total_frames, dst=0):
    """Frame-sharded output hand-off: masks uint8 [n, t_local, H, W] of this rank's frames -> on rank `dst` the masks of ALL
    frames [n, total_frames, H, W] (device tensor), None on the other ranks (10 x t_local x H x W bytes
    per rank, 9.2 MB per 720p frame).  Without a process group: the input."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return masks
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [len(inference_shard(total_frames, r, world)) for r in range(world)]
    tmax = max(sizes)
    n = masks.shape[0]
    dev = masks.device
    staged = dist.get_backend() == "gloo"
    src = masks.detach().cpu() if staged else masks
    pad = src.new_zeros((tmax, n) + tuple(masks.shape[2:]))
    pad[: masks.shape[1]] = src.transpose(0, 1)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    full = torch.cat([b[:k] for b, k in zip(bufs, sizes)], dim=0)       # [T, n, H, W]
    return full.transpose(0, 1).contiguous().to(dev)


def all_gather_rows(t):
    """Equal-sized flat blocks, one per rank -> [world, n] in rank order (the per-layer exchange of the split-KV video decoder: one
    109 KB flash partial per rank).  On RCCL the collective is queued behind the producer on the caller's stream and the
    consumer (ops.attention_merge) behind it: nothing overlaps it by construction -- the next kernel needs the merged rows -- so there
    is no side stream here.  Without a process group: t[None]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t.reshape(1, -1)
    world = dist.get_world_size()
    flat = t.reshape(-1).contiguous()
    if dist.get_backend() == "gloo":                            # CPU tests / one-GPU test rigs: staged through the host
        out = torch.empty((world * flat.numel(),), dtype=flat.dtype)
        dist.all_gather_into_tensor(out, flat.detach().cpu())
        return out.view(world, -1).to(t.device)
    out = flat.new_empty((world * flat.numel(),))
    dist.all_gather_into_tensor(out, flat)
    return out.view(world, -1)


def all_reduce_sum(t):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "gloo" and t.is_cuda:          # test rigs without RCCL: stage through the host
            h = t.cpu().contiguous()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            return h.to(t.device)
        t = t.contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def warm_up(device):
    """One small collective of every kind the frame-sharded path uses (all_gather_into_tensor on the side stream, all_reduce,
    gather, and the scalar MAX of the timing), so that RCCL's communicator / channel set-up happens here and not inside a timed
    step.  A no-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size()
    dev = torch.device(device)
    t_local = len(inference_shard(2 * world + 1, dist.get_rank(), world))          # ragged on purpose: exercises the padding
    x = torch.full((t_local, 4, 8), float(dist.get_rank()), device=dev)
    g = all_gather_frames(x, 2 * world + 1)
    assert g.shape[0] == 2 * world + 1
    all_reduce_sum(torch.ones(16, device=dev))
    assert all_gather_rows(torch.full((32,), float(dist.get_rank()), device=dev)).shape == (world, 32)
    gather_frame_masks(torch.zeros((2, t_local, 4, 4), dtype=torch.uint8, device=dev), 2 * world + 1)
    max_over_ranks(0.0, "cpu" if dist.get_backend() == "gloo" else dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    dist.barrier()
