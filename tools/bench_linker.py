"""Times ovis_hungarian_link on 36 frames x 100 queries: well-separated embeddings and the near-degenerate ones of random-init weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops
g = torch.Generator().manual_seed(0)
T, Q, C = 36, 100, 256
base = torch.randn(Q, C, generator=g)
cases = {"separated": torch.stack([base[torch.randperm(Q, generator=g)] + 0.3 * torch.randn(Q, C, generator=g) for _ in range(T)]),
         "near-degenerate": torch.stack([torch.randn(1, C, generator=g) + 0.02 * torch.randn(Q, C, generator=g) for _ in range(T)])}
for name, emb in cases.items():
    e = emb.cuda()
    for _ in range(3):
        ops.hungarian_link(e)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        ops.hungarian_link(e)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per 36-frame chain")
