"""Encoder-layer projections (5 x 19 320 rows, C = 256): add + value GEMM + offset / weight GEMM against the two-output GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator().manual_seed(0)
for T in (5, 36):
    S, C, N2 = 19320, 256, 288
    src = torch.randn(T, S, C, generator=g).cuda(); pos = torch.randn(S, C, generator=g).cuda()
    w = (torch.randn(C + N2, C, generator=g) / 16).cuda(); b = torch.randn(C + N2, generator=g).cuda()
    wv, wo, bv, bo = w[:C].contiguous(), w[C:].contiguous(), b[:C].contiguous(), b[C:].contiguous()
    ops.set_f32_gemm_mode(3); ops.f16x2_begin("cuda")
    posw = ops.gemm_nt(pos, wo, None, cw=True)
    t_add = timeit(lambda: ops.add_bcast(src, pos))
    q = ops.add_bcast(src, pos)
    t_v = timeit(lambda: ops.gemm_nt(src, wv, bv, cw=True))
    t_o = timeit(lambda: ops.gemm_nt(q, wo, bo, cw=True))
    t_d = timeit(lambda: ops.gemm_nt_dual(src, w, b, posw, C))
    print(f"T={T}: add {t_add:.1f} + value {t_v:.1f} + offsets/weights {t_o:.1f} = {t_add + t_v + t_o:.1f} us;  two-output GEMM {t_d:.1f} us", flush=True)
