import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops
from tools.bench_gemm import timeit
for (Q, K) in [(100, 482), (100, 1196), (200, 1203)]:
    probs = torch.rand(Q, K, device="cuda").softmax(-1)
    rows = torch.arange(Q, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: ops.topk_entropy(probs, rows, 10), n=20)
    print(f"Q={Q} K={K}: {ms*1e3:.1f} us")
