"""Premise check for a plane-walking fp16x2 mode: the plain fp16 ping-pong GEMM on a 3 K axis (what hi / lo planes would run as) against
the register-splitting f32-A fp16x2 kernel, encoder shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator().manual_seed(0)
M = 96600
ops.set_f32_gemm_mode(3); ops.f16x2_begin("cuda")
for name, N, K in (("value/out-proj", 256, 256), ("dual", 544, 256), ("ffn1", 1024, 256), ("ffn2", 256, 1024)):
    a = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    t_fh = timeit(lambda: ops.gemm_nt(a, w, b, None, 0, cw=True))
    a16 = torch.randn(M, 3 * K, generator=g).half().cuda(); w16 = (torch.randn(N, 3 * K, generator=g) / K ** 0.5).half().cuda()
    t_16 = timeit(lambda: ops.gemm_nt_f16(a16, w16, b, None, 0, out_f16=False))
    t_split = timeit(lambda: ops.cast_f16(a))
    print(f"{name:14s} N={N} K={K}: fp16x2 (f32 A) {t_fh:.1f} us | fp16 GEMM on 3K {t_16:.1f} us | one pass over A (cast) {t_split:.1f} us", flush=True)
