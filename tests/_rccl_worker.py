"""Worker of tests/test_sharded_gpu.py::test_rccl_collectives_of_the_frame_sharded_path: one rank per GPU over RCCL (backend "nccl"),
started by torch.distributed.run.  Exercises exactly the branches of openvis_amd/distributed.py that gloo runs never take: the async
all_gather_into_tensor on the side stream, the device all-reduce and dist.gather of device tensors."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvis_amd import distributed as D


def main():
    # OVIS_RCCL_TEST_BACKEND=gloo (tests/test_distributed_cpu.py): the same script on CPU tensors, to check its own logic where no
    # second GPU exists; the GPU test runs it with nccl
    backend = os.environ.get("OVIS_RCCL_TEST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))    # before any other HIP call
    rank, world, local_rank = D.init_from_env(backend)
    import torch.distributed as dist
    assert dist.get_backend() == backend and world >= 2
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    D.warm_up(dev)
    T, Q, C = 4 * world + 3, 100, 256                            # ragged shards (the C4 layout: 36 frames over 8 ranks -> 5,5,5,5,4,4,4,4)
    full = torch.arange(T * Q * C, dtype=torch.float32).view(T, Q, C) * 1e-3
    mine = D.inference_shard(T, rank, world)
    local = full[mine.start:mine.stop].to(dev)
    h = D.all_gather_frames_async(local, T)                      # side stream; compute continues on the current stream meanwhile
    busy = torch.ones(1024, 1024, device=dev) @ torch.ones(1024, 1024, device=dev)
    got = h.wait()
    assert got.device.type == dev.type and torch.equal(got.cpu(), full), "async all-gather of query embeddings"
    assert float(busy[0, 0]) == 1024.0
    logits = torch.full((Q, 483), float(rank + 1), device=dev)
    s = D.all_reduce_sum(logits)
    assert torch.equal(s.cpu(), torch.full((Q, 483), float(world * (world + 1) // 2))), "logit all-reduce"
    masks = torch.zeros((10, len(mine), 8, 16), dtype=torch.uint8, device=dev)
    for j, t in enumerate(mine):
        masks[:, j] = t % 251
    out = D.gather_frame_masks(masks, T, dst=0)
    if rank == 0:
        assert out.shape == (10, T, 8, 16) and all(int(out[:, t].min()) == int(out[:, t].max()) == t % 251 for t in range(T)), "mask gather"
    else:
        assert out is None
    # the per-layer exchange of the split-KV video decoder: one equal-sized packed block per rank, stacked in rank order
    rows = D.all_gather_rows(torch.arange(27300, dtype=torch.float32, device=dev) + 1e5 * rank)
    assert rows.shape == (world, 27300) and rows.device.type == dev.type
    assert all(torch.equal(rows[r].cpu(), torch.arange(27300, dtype=torch.float32) + 1e5 * r) for r in range(world)), "all_gather_rows"
    assert D.max_over_ranks(float(rank), dev) == float(world - 1)
    D.barrier()
    if rank == 0:
        print("RCCL_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
