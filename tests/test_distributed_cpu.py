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


def _worker_c4(rank, world, port, T, out):
    """One rank of the C4 layout (BASELINE.json configs[3]): the control flow of BriVIS.forward(frame_range=...) with the
    per-frame GPU work replaced by a deterministic function of the frame index (openvis_amd/brivis.py:47-84)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from openvis_amd import distributed as D
    D.init_from_env("gloo")
    assert D.world_size() == world
    Q, C, K1, H, W, n = 6, 8, 5, 4, 6, 3
    g = torch.Generator().manual_seed(0)
    embeds = torch.randn(T, Q, C, generator=g)                   # what every frame's decoder would produce
    logits = torch.randn(T, Q, K1, generator=g)
    masks = (torch.rand(n, T, H, W, generator=g) > 0.5).to(torch.uint8)
    mine = D.inference_shard(T, rank, world)
    b0, b1 = mine.start, mine.stop
    h = D.all_gather_frames_async(embeds[b0:b1].clone(), T)      # starts "on the side stream"
    local = logits[b0:b1].mean(0) * (float(b1 - b0) / float(T))  # classify_sharded: weighted local mean ...
    emb = h.wait()
    total = D.all_reduce_sum(local.clone())                      # ... all-reduced = mean over ALL frames
    full = D.gather_frame_masks(masks[:, b0:b1].contiguous(), T, dst=0)
    ok_masks = (full is None) if rank != 0 else bool(torch.equal(full, masks))
    out.put((rank, b1 - b0, bool(torch.equal(emb, embeds)), float((total - logits.mean(0)).abs().max()), ok_masks))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_c4_layout_world_size_8_end_to_end():
    """36 frames over 8 ranks = 5,5,5,5,4,4,4,4: padded all-gather of the query embeddings gives every
    rank the full sequence in frame order, the weighted logit all-reduce equals the mean over all 36 frames, and the
    selected masks of all frames arrive on rank 0 in frame order."""
    world, T = 8, 36
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c4, args=(r, world, port, T, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [r[1] for r in res] == [5, 5, 5, 5, 4, 4, 4, 4]
    assert all(r[2] for r in res)
    assert all(r[3] < 1e-6 for r in res)
    assert all(r[4] for r in res)


def _worker_warm_up(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from openvis_amd import distributed as D
    D.init_from_env("gloo")
    D.warm_up("cpu")                                   # every collective kind of the frame-sharded path, ragged shards
    out.put((rank, D.world_size()))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_communicator_warm_up_runs_every_collective_kind():
    """bench.py calls distributed.warm_up before the timed region for world > 1 (RCCL builds rings at the first collective)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_warm_up, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs) and res == [(0, 2), (1, 2)]
    from openvis_amd import distributed as D
    D.warm_up("cpu")                                   # no process group: a no-op


def test_rccl_worker_logic_on_gloo(tmp_path):
    """tests/_rccl_worker.py (the script the >= 2-GPU RCCL test launches, one rank per GPU) run here with gloo on CPU tensors, 3 ranks:
    ragged shards, gather order, all-reduce value and the mask gather are what the script asserts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OVIS_RCCL_TEST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tests", "_rccl_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0 and "RCCL_OK world=3" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def _forced_one_rank(port, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from openvis_amd import distributed as D
    assert D.init_from_env("gloo") == (0, 1, 0) and not dist.is_initialized()           # world 1: a no-op ...
    assert D.backend_name() is None
    assert D.init_from_env("gloo", force=True) == (0, 1, 0) and dist.is_initialized()   # ... unless forced (round 6: the 1-rank RCCL group)
    assert D.backend_name() == "gloo" and D.world_size() == 1
    x = torch.arange(7 * 4 * 8, dtype=torch.float32).view(7, 4, 8)
    assert torch.equal(D.all_gather_frames(x, 7), x)
    assert torch.equal(D.all_reduce_sum(torch.ones(5)), torch.ones(5))
    m = (torch.arange(2 * 7 * 3 * 3) % 2).to(torch.uint8).view(2, 7, 3, 3)
    assert torch.equal(D.gather_frame_masks(m, 7, dst=0), m)
    assert D.max_over_ranks(1.5) == 1.5
    D.warm_up("cpu")
    dist.destroy_process_group()
    q.put("ok")


def test_forced_one_rank_process_group():
    """distributed.init_from_env(force=True): a ONE-rank group (what the GPU suite uses to run the RCCL branches on a single device) -- here on
    gloo: every helper of the frame-sharded path is the identity at world 1, and without `force` no group is created."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = ctx.Process(target=_forced_one_rank, args=(port, q))
    p.start(); p.join(120)
    assert p.exitcode == 0 and q.get(timeout=5) == "ok"


def test_bench_side_guard_prints_the_line_once_and_leaves_when_a_side_measurement_hangs():
    """bench.py's multi-rank side measurements (frame_sharded, split_clip) run collectives after the headline is complete; _SideGuard is what
    keeps a hung collective from costing the JSON line: past the deadline rank 0 prints the line with an error in that field, exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "line = {'metric': 'm', 'value': 1.5, 'split_clip': None}\n"
            "g = bench._SideGuard(line, 0.5)\n"
            "g.arm('frame_sharded'); g.disarm()\n"                  # a side measurement that returns in time: nothing happens
            "g.arm('split_clip'); time.sleep(30)\n"                  # one that hangs
            "print('NOT REACHED')\n") % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["value"] == 1.5 and "no result within" in d["split_clip"]["error"] and "split_clip" in p.stderr
    # ... and a guard that was emitted normally does not print again when a late timer fires
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "g = bench._SideGuard({'value': 2}, 0.3); g.arm('x'); g.emit(); time.sleep(5)\n") % root
    p = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and [ln for ln in p.stdout.splitlines() if ln.strip()] == ['{"value": 2}'], (p.stdout, p.stderr[-500:])
