"""Does v_mfma_f32_*_f16 keep fp16 subnormal INPUTS on gfx950?  And does the in-register fp16 split (v_fma_mix*_f16) produce them?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

dev = "cuda"
M, N, K = 70000, 256, 256
# (1) fp16 MFMA kernel on subnormal inputs: a = 2^-20 (subnormal in fp16), w = 1 -> K * 2^-20
a = torch.full((M, K), 2.0 ** -20, dtype=torch.float16, device=dev)
w = torch.ones((N, K), dtype=torch.float16, device=dev)
out = ops.gemm_nt_f16(a, w)
print("fp16 MFMA, subnormal A:", out[0, 0].item(), "expected", K * 2.0 ** -20)
a = torch.ones((M, K), dtype=torch.float16, device=dev)
w = torch.full((N, K), 2.0 ** -20, dtype=torch.float16, device=dev)
out = ops.gemm_nt_f16(a, w)
print("fp16 MFMA, subnormal B:", out[0, 0].item(), "expected", K * 2.0 ** -20)
# (2) the fp16x2 path on values whose lo is subnormal: a = 1/16 + 2^-24/16 ... choose a*16 = 1 + 2^-13 -> hi = 1, lo = 2^-13 (normal);
#     a*16 = 2^-3 + 2^-18 -> hi = 2^-3, lo = 2^-18 (subnormal fp16)
ops.set_f32_gemm_mode(3)
ops.f16x2_begin(dev)
for v16, name in ((1.0 + 2.0 ** -13, "lo normal (2^-13)"), (2.0 ** -3 + 2.0 ** -18, "lo subnormal (2^-18)")):
    a = torch.full((96600, 256), v16 / 16.0, device=dev)
    w = torch.ones((256, 256), device=dev)
    out = ops.gemm_nt(a, w, cw=True)
    exact = 256 * (v16 / 16.0)
    print(f"fp16x2 {name}: got {out[0,0].item():.10e} exact {exact:.10e} hi-only {256 * (torch.tensor(v16).half().item()) / 16:.10e}")
# weight side: w * scale with subnormal lo
wv = 1.0 + 2.0 ** -13
a = torch.ones((96600, 256), device=dev)
for wval, name in ((1.0 + 2.0 ** -13, "w lo normal"), (1.0, "w plain")):
    w = torch.full((256, 256), wval, device=dev)
    w[0, 0] = 2.0 ** 14 * 1.5          # pins the scale at 1: planes hold w itself
    w[1, :] = 2.0 ** -3 + 2.0 ** -18   # row 1: lo subnormal
    h2, s = ops.h2_of(w)
    out = ops.gemm_nt(a, w, cw=True)
    print(f"{name}: scale {s}; row1 lo plane value {h2[1,1,5].item():.6e}; out[0,1] {out[0,1].item():.10e} exact {256 * (2.0**-3 + 2.0**-18):.10e}")
ops.set_f32_gemm_mode(1)
