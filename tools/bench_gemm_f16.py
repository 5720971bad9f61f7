"""fp16 MFMA GEMM micro-benchmark at the CLIP ViT shapes (TFLOP/s), next to torch/hipBLASLt for reference."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

if __name__ == "__main__":
    dev = "cuda:0"
    only_ours = len(sys.argv) > 1 and sys.argv[1] == "ours"
    for (M, N, K) in [(98500, 2304, 768), (98500, 768, 768), (98500, 3072, 768), (98500, 768, 3072), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) / K ** 0.5).half(); b = torch.randn(N, device=dev)
        ms = timeit(lambda: ops.gemm_nt_f16(a, w, b, None, 0, out_f16=True))
        rec = {"gemm_f16": [M, N, K], "ms": round(ms, 4), "TF": round(2 * M * N * K / ms / 1e9, 1)}
        if not only_ours:
            bh = b.half()
            ms_t = timeit(lambda: torch.nn.functional.linear(a, w, bh))
            rec.update({"torch_ms": round(ms_t, 4), "torch_TF": round(2 * M * N * K / ms_t / 1e9, 1)})
        print(json.dumps(rec))
