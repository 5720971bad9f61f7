"""Lab: does the row stride of A pace the f32-A (fp16x2) K loop?  One GEMM (N = 256, K = 256, M = 96 600, 192-row tiles), A a column slice of a
wider matrix: lda = 256 (dense), 1024, 4096 floats.  K loop per tile from the in-kernel stamps."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openvis_amd import ops, _lib
from openvis_amd.ops import _ll, h2_of

M, N, K = 96600, 256, 256
g = torch.Generator().manual_seed(0)
ops.set_f32_gemm_mode(3); ops.f16x2_begin("cuda")
w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
h2, ws = h2_of(w)
out = torch.empty(M, N, device="cuda")
stamps = torch.zeros(65536, dtype=torch.int64, device="cuda")
for rnd in range(2):
    for lda in (256, 1024, 4096):
        big = torch.randn(M, lda, generator=g).cuda()
        f = lambda: _lib.call("ovis_gemm_nt_f32_h2", big, _ll(lda), w, _ll(K), h2, _ll(w.numel()), ctypes.c_float(ws), out, _ll(N), M, N, K, b, None, _ll(N), 0,
                              _lib.stream_ptr())
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        stamps.zero_(); _lib.call("ovis_pp_debug", 0, stamps); f(); torch.cuda.synchronize(); _lib.call("ovis_pp_debug", 0, None)
        st = stamps[:256 * 16 * 2 * 4].cpu().numpy().reshape(256, 16, 2, 4)
        ok = st[:, :, 0, 2] > 0
        kl = (st[:, :, 0, 1] - st[:, :, 0, 0])[ok] * 0.01
        print(f"lda = {lda:5d} floats: launch {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, K loop per tile median {np.median(kl):.2f} us (p10 {np.percentile(kl, 10):.2f}, p90 {np.percentile(kl, 90):.2f}) = {np.median(kl) / 8:.2f} us per K step", flush=True)
        del big
