"""Lab: in-kernel stamps of the fp16x2 (f32-A) ping-pong GEMM on the encoder shapes -- K loop and epilogue time per tile
(ovis_pp_debug; stamps = s_memrealtime ticks of 10 ns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openvis_amd import _lib
if os.environ.get("OVIS_LAB_LIB"):
    _lib.LIB_PATH = os.environ["OVIS_LAB_LIB"]      # lab only: a variant build of the library
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
stamps = torch.zeros(65536, dtype=torch.int64, device="cuda")
for name, N, K, act in (("out-proj", 256, 256, 0), ("ffn1", 1024, 256, 1), ("ffn2", 256, 1024, 0), ("fp16 qkv", 2304, 768, 0)):
    if name.startswith("fp16"):
        a = torch.randn(98500, K, generator=g).half().cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda(); b = torch.randn(N, generator=g).cuda()
        f = lambda: ops.gemm_nt_f16(a, w, b, None, 0, out_f16=True)
    else:
        a = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
        f = lambda: ops.gemm_nt(a, w, b, None, act, cw=True)
    import hashlib
    digest = hashlib.sha1(f().cpu().numpy().tobytes()).hexdigest()[:12]
    t = timeit(f)
    stamps.zero_()
    _lib.call("ovis_pp_debug", 0, stamps)
    f(); torch.cuda.synchronize()
    _lib.call("ovis_pp_debug", 0, None)
    st = stamps[:256 * 16 * 2 * 4].cpu().numpy().reshape(256, 16, 2, 4)
    ok = st[:, :, 0, 2] > 0
    kl = (st[:, :, 0, 1] - st[:, :, 0, 0])[ok] * 0.01
    ep = (st[:, :, 0, 2] - st[:, :, 0, 1])[ok] * 0.01
    nxt = (st[:, 1:, 0, 0] - st[:, :-1, 0, 0])[ok[:, 1:] & ok[:, :-1]] * 0.01
    tiles = ok.sum(1)
    if not ok.any():
        print(f"{name:10s} N={N} K={K}: launch {t:.1f} us | sha1 {digest} | no stamps"); continue
    span = (st[:, :, 0, 2].max() - st[:, 0, 0, 0][st[:, 0, 0, 0] > 0].min()) * 0.01
    print(f"{name:10s} N={N} K={K}: launch {t:.1f} us | sha1 {digest} | tiles per workgroup {tiles.min()}-{tiles.max()} | K loop median {np.median(kl):.2f} us (p10 {np.percentile(kl,10):.2f}, p90 {np.percentile(kl,90):.2f})"
          f" | epilogue median {np.median(ep):.2f} us (p10 {np.percentile(ep,10):.2f}, p90 {np.percentile(ep,90):.2f}) | tile period median {np.median(nxt):.2f} | first start -> last end {span:.1f} us", flush=True)
