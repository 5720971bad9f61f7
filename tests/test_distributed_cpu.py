"""CPU, world_size 2 over gloo: the clip-sharding / timing plumbing bench.py uses for --gpus N."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from openvis_amd import distributed as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    mine = list(D.inference_shard(total, rank, world))
    D.barrier()
    elapsed = 1.0 + rank                      # rank 1 is "slower"
    tmax = D.max_over_ranks(elapsed)
    nsum = D.sum_over_ranks(len(mine))
    out.put((rank, mine, tmax, nsum))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world, total = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    shards = [r[1] for r in res]
    assert shards[0] == [0, 1, 2, 3] and shards[1] == [4, 5, 6]          # contiguous, InferenceSampler layout
    assert sorted(sum(shards, [])) == list(range(total))
    assert all(abs(r[2] - 2.0) < 1e-12 for r in res)                     # MAX over ranks
    assert all(abs(r[3] - total) < 1e-12 for r in res)


def _worker_gather(rank, world, port, T, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from openvis_amd import distributed as D
    D.init_from_env("gloo")
    full = torch.arange(T * 3 * 4, dtype=torch.float32).view(T, 3, 4)          # "pred_embeds" [T,Q,C]
    mine = D.inference_shard(T, rank, world)
    gathered = D.all_gather_frames(full[mine.start:mine.stop].clone(), T)     # C5: one all-gather of padded blocks
    logits = torch.full((2, 5), float(rank + 1)) * len(mine)                  # per-rank sum over local frames
    total = D.all_reduce_sum(logits.clone())
    out.put((rank, torch.equal(gathered, full), total[0, 0].item()))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_frame_sharded_all_gather_and_logit_all_reduce():
    """BriVIS frame sharding: uneven shards 3/2 of T=5 are padded, gathered and re-assembled in order."""
    world, T = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gather, args=(r, world, port, T, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(r[1] for r in res)
    assert all(abs(r[2] - (1 * 3 + 2 * 2)) < 1e-9 for r in res)              # rank0: 3 frames x 1, rank1: 2 frames x 2


def test_shard_edge_cases():
    from openvis_amd.distributed import inference_shard
    assert list(inference_shard(0, 0, 4)) == []
    assert [len(inference_shard(3, r, 8)) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    assert [list(inference_shard(36, r, 8)) for r in (0, 7)] == [[0, 1, 2, 3, 4], [32, 33, 34, 35]]
