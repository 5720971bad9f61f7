"""bf16x3 f32 GEMM time against K at the pixel-decoder shape (M = 96 600, N = 256): separates the per-tile fixed cost
(prologue + epilogue) from the per-K-tile cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops
from tools.bench_gemm import timeit
M = 96600
for N in (256, 1024):
    for K in (128, 256, 512, 1024, 2048):
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
        ms = timeit(lambda: ops.gemm_nt(a, w, b, None, 0, cw=True), n=10)
        print(f"N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TF(f32-equiv)")
