"""LayerNorm-folded fp16 GEMMs of the CLIP tower (in_proj, c_fc + QuickGELU) on 256-row against 192-row tiles (round 6: the folded
instantiations take 192-row tiles; lab switch ovis_pp_tile_rows), M = 500 crops x 197 tokens.  Equality of the two outputs is checked
(the same dot products per element in the same order: tile height does not enter the arithmetic).
Needs the LAB build described in profiles/r06/tile_rows.txt: the shipped launcher keeps the folded instantiations on 256-row tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib
from tools.bench_dual import timeit

g = torch.Generator().manual_seed(0)
M, C = 500 * 197, 768
x = torch.randn(M, C, generator=g).half().cuda()
stats = ops.row_stats_f16(x)
for name, N, act in (("in_proj [98500, 2304, 768]", 2304, ops.ACT_NONE), ("c_fc [98500, 3072, 768] + QuickGELU", 3072, ops.ACT_QUICKGELU)):
    w = (torch.randn(N, C, generator=g) / C ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda(); beta = (0.05 * torch.randn(C, generator=g)).cuda()
    wg, s, c = ops.fold_layernorm(w, b, gamma, beta)
    outs, row = {}, []
    for tm in (256, 192, 256, 192, 0):
        _lib.call("ovis_pp_tile_rows", tm)
        outs[tm] = ops.gemm_nt_f16_ln(x, wg, s, c, stats, act)
        row.append(f"tm {tm}: {timeit(lambda: ops.gemm_nt_f16_ln(x, wg, s, c, stats, act)):.1f} us")
    _lib.call("ovis_pp_tile_rows", 0)
    print(name, " | ".join(row), "| 192 == 256:", bool(torch.equal(outs[192], outs[256])), "| auto == 192:", bool(torch.equal(outs[0], outs[192])), flush=True)
