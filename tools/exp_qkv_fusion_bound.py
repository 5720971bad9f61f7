"""Measured pieces for the QKV-projection -> attention fusion question (round-4 review, item 6): the QKV GEMM of the CLIP tower on 256-row and
on 192-row tiles (what a smaller tile costs on the shipped kernel), and the attention kernel whole / staging only / compute on resident LDS."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib
from tools.bench_gemm_f16 import timeit

M, N, K = 500 * 197, 2304, 768
a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half(); b = torch.randn(N, device="cuda")
res = {}
for rounds in range(3):
    for tm in (256, 192):
        _lib.call("ovis_pp_tile_rows", tm)
        ms = timeit(lambda: ops.gemm_nt_f16(a, w, b, None, 0, out_f16=True), n=10)
        res.setdefault(tm, []).append(ms)
_lib.call("ovis_pp_tile_rows", 0)
for tm, v in res.items():
    print(f"QKV GEMM M={M} N={N} K={K}, {tm}-row tiles: {min(v)*1e3:.1f} us min, {sorted(v)[1]*1e3:.1f} us median  ({2*M*N*K/min(v)/1e9:.0f} TF)")
B, H, L, D = 500, 12, 197, 64
C = H * D
qkv = torch.randn(B * L, 3 * C, device="cuda").half()
f = lambda: ops.attention_f16(qkv, qkv[:, C:], qkv[:, 2 * C:], B, H, L, L, D, L * 3 * C, 3 * C, L * 3 * C, 3 * C, L * 3 * C, 3 * C)
for dbg, name in ((0, "whole"), (1, "K / V staging only"), (2, "compute on resident LDS only")):
    _lib.call("ovis_attention_f16_debug", dbg)
    t = [timeit(f, n=20) for _ in range(3)]
    print(f"attention {B} crops x {H} heads x {L} tokens, {name}: {min(t)*1e3:.1f} us")
_lib.call("ovis_attention_f16_debug", 0)
