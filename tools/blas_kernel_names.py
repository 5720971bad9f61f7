"""Which hipBLASLt (Tensile) kernels torch picks for the CLIP ViT GEMM shapes: run under `rocprofv3 --kernel-trace --stats` and read the
kernel names (they spell the macro tile, the wave layout and the LDS / prefetch options).  Evidence only; the product never calls it."""
import torch
dev = "cuda:0"
for (M, N, K) in [(98500, 2304, 768), (98500, 768, 768), (98500, 3072, 768), (98500, 768, 3072), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) / K ** 0.5).half(); b = torch.randn(N, device=dev).half()
    for _ in range(5):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
