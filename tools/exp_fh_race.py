"""Run-to-run determinism of the fp16x2 ping-pong variants; prints where outputs differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

M, N, K = 96600, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(M + K)
a = torch.randn(M, K, generator=g).cuda()
w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
b = (0.1 * torch.randn(N, generator=g)).cuda()
r = (2 * torch.randn(M, N, generator=g) + torch.randn(M, 1, generator=g)).cuda()
gamma, beta = (1 + 0.3 * torch.randn(N, generator=g)).cuda(), (0.2 * torch.randn(N, generator=g)).cuda()
from openvis_amd import _lib
DBG = int(os.environ.get("PP_DBG", "0"))
_lib.lib().ovis_pp_debug(DBG, None)
print("dbg", DBG)
for mode in (2, 3):
    ops.set_f32_gemm_mode(mode)
    if mode == 3:
        ops.f16x2_begin("cuda")
    for name, fn in (("plain", lambda: ops.gemm_nt(a, w, b, None, 0, cw=True)), ("res", lambda: ops.gemm_nt(a, w, b, r, 0, cw=True)),
                     ("LN", lambda: ops.gemm_nt_layernorm(a, w, b, r, gamma, beta))):
        ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(10):
            o = fn()
            d = (o != ref)
            if d.any():
                bad += 1
                if bad == 1:
                    rows = d.any(1).nonzero().flatten()
                    cols = d.any(0).nonzero().flatten()
                    print(f"mode {mode} {name}: it {it}: {int(d.sum())} elements differ, rows {rows[:12].tolist()} (n={len(rows)}) rows%192 {sorted(set((rows % 192).tolist()))[:24]} "
                          f"cols {cols[:12].tolist()} (n={len(cols)}), max |diff| {(o - ref).abs().max().item():.3e}")
        print(f"mode {mode} {name}: {bad} of 10 repeats differ")
ops.set_f32_gemm_mode(1)
