"""192-row against 256-row tiles of the ping-pong GEMM on the pixel decoder's fp16x2 shapes (5 x 19 320 rows, C = 256): the two-output
encoder GEMM (N = 544), ffn1 (N = 1024, ReLU), and a plain 256-column projection.  Lab switch ovis_pp_tile_rows (0 = automatic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib
from tools.bench_dual import timeit

g = torch.Generator().manual_seed(0)
T, S, C, N2 = 5, 19320, 256, 288
src = torch.randn(T, S, C, generator=g).cuda(); pos = torch.randn(S, C, generator=g).cuda()
w = (torch.randn(C + N2, C, generator=g) / 16).cuda(); b = torch.randn(C + N2, generator=g).cuda()
w1 = (torch.randn(1024, C, generator=g) / 16).cuda(); b1 = torch.randn(1024, generator=g).cuda()
ops.set_f32_gemm_mode(3); ops.f16x2_begin("cuda")
posw = ops.gemm_nt(pos, w[C:].contiguous(), None, cw=True)
x2 = src.view(-1, C)
for name, fn in (("two-output encoder GEMM [96600, 544, 256]", lambda: ops.gemm_nt_dual(src, w, b, posw, C)),
                 ("ffn1 [96600, 1024, 256] + ReLU", lambda: ops.gemm_nt(x2, w1, b1, None, ops.ACT_RELU, cw=True)),
                 ("value-sized projection [96600, 256, 256]", lambda: ops.gemm_nt(x2, w[:C].contiguous(), b[:C].contiguous(), cw=True))):
    row = []
    for tm in (0, 256, 192, 256, 192):
        _lib.call("ovis_pp_tile_rows", tm)
        row.append(f"tm {tm}: {timeit(fn):.1f} us")
    _lib.call("ovis_pp_tile_rows", 0)
    print(name, " | ".join(row), flush=True)
