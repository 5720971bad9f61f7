"""Multi-GPU plumbing for the clip-sharded eval path (one process per GPU, torch.distributed; "nccl" == RCCL on ROCm).

OpenVIS' offline decoder attends over every frame of a clip, so the unit of parallelism is the CLIP: ranks own
contiguous shards of the clip list exactly like detectron2's InferenceSampler (openvis/data/build.py:238-247) and
there is no collective on the data path.  The only collective is the scalar MAX used for timing."""
import os

import torch


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run). Returns
    (rank, world, local_rank); a no-op for world == 1."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
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


def all_gather_frames(local, total_frames):
    """All-gather of per-frame rows over contiguous frame shards: `local` is this rank's
    [t_local, ...] block of a [total_frames, ...] tensor sharded with `inference_shard`; blocks are padded to the
    largest shard so ONE all_gather moves everything (<= 512 KB per rank for [5,100,256] f32 — latency-bound)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    sizes = [len(inference_shard(total_frames, r, world)) for r in range(world)]
    tmax = max(sizes)
    pad = local.new_zeros((tmax,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:          # test rigs without RCCL: stage through the host
        pad = pad.cpu()
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0).to(dev)


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
