"""Epilogue cost of the CLIP out-proj shape (M = 475 x 197, N = K = 768): f32 out + residual / f32 out / f16 out, plus a
plain residual-add copy kernel of the same bytes for reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

M, N, K = 475 * 197, 768, 768
a = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
b = torch.randn(N, device="cuda")
r = torch.randn(M, N, device="cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("f32 out + residual : %.1f us" % t(lambda: ops.gemm_nt_f16(a, w, b, r, 0, out_f16=False)))
print("f32 out            : %.1f us" % t(lambda: ops.gemm_nt_f16(a, w, b, None, 0, out_f16=False)))
print("f32 out, no bias   : %.1f us" % t(lambda: ops.gemm_nt_f16(a, w, None, None, 0, out_f16=False)))
print("f16 out            : %.1f us" % t(lambda: ops.gemm_nt_f16(a, w, b, None, 0, out_f16=True)))
y = torch.empty_like(r)
print("torch add (r + r -> y, 3 x 287 MB): %.1f us" % t(lambda: torch.add(r, r, out=y)))
print("torch copy (r -> y, 2 x 287 MB)   : %.1f us" % t(lambda: y.copy_(r)))
