"""Times the CLIP ViT-B/16 GEMM shapes of the bench workload (M = 475 crops x 197 tokens) through ovis_gemm_nt_f16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

M = 475 * 197
for (N, K, of16) in [(768, 768, False), (768, 3072, False), (2304, 768, True), (3072, 768, True)]:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if not of16 else None
    for _ in range(3):
        ops.gemm_nt_f16(a, w, b, r, 0, out_f16=of16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_nt_f16(a, w, b, r, 0, out_f16=of16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"N={N} K={K} f16out={of16}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TF")
