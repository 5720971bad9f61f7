"""The decoder's small launches alone: ln_mlp3 (decoder_norm + mask MLP), the split-K FFN2 GEMM, FFN1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator().manual_seed(0)
ops.set_f32_gemm_mode(3); ops.f16x2_begin("cuda")
for M in (100, 500):
    C = 256
    x = torch.randn(M, C, generator=g).cuda(); gm, bt = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    ws = [(torch.randn(C, C, generator=g) / 16).cuda() for _ in range(3)]; wts = [w.t().contiguous() for w in ws]
    bs = [torch.randn(C, generator=g).cuda() for _ in range(3)]
    t_f = timeit(lambda: ops.ln_mlp3(x, gm, bt, wts, bs))
    def sep():
        d = ops.layernorm(x, gm, bt)
        h = ops.gemm_nt(d, ws[0], bs[0], None, ops.ACT_RELU, cw=True); h = ops.gemm_nt(h, ws[1], bs[1], None, ops.ACT_RELU, cw=True)
        return ops.gemm_nt(h, ws[2], bs[2], cw=True)
    t_s = timeit(sep)
    w1 = (torch.randn(2048, C, generator=g) / 16).cuda(); w2 = (torch.randn(C, 2048, generator=g) / 45).cuda()
    b1, b2 = torch.randn(2048, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    h = ops.gemm_nt(x, w1, b1, None, ops.ACT_RELU, cw=True)
    t1 = timeit(lambda: ops.gemm_nt(x, w1, b1, None, ops.ACT_RELU, cw=True))
    t2 = timeit(lambda: ops.gemm_nt(h, w2, b2, x, cw=True))
    _lib.call("ovis_set_skinny_gemm", 2)
    t2o = timeit(lambda: ops.gemm_nt(h, w2, b2, x, cw=True)); ref = ops.gemm_nt(h, w2, b2, x, cw=True)
    _lib.call("ovis_set_skinny_gemm", 1)
    same = torch.equal(ref, ops.gemm_nt(h, w2, b2, x, cw=True))
    print(f"M={M}: ln_mlp3 {t_f:.1f} us (LayerNorm + 3 GEMMs {t_s:.1f});  FFN1 {t1:.1f}, FFN2 {t2:.1f} us (128-row split-K form {t2o:.1f}; same bits {same})", flush=True)
