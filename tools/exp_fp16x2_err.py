"""Where does the fp16x2 error come from?  GPU result vs f64 reference vs an f64 emulation of the same split (RNE) on the test's data."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

M, N, K = 96600, 256, 256
g = torch.Generator().manual_seed(M + K)
a = (torch.randn(M, K, generator=g) * torch.exp((1.5 * torch.randn(M, 1, generator=g)).clamp(-4, 4))).cuda()
w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
b = torch.randn(N, generator=g).cuda()
r = torch.randn(M, N, generator=g).cuda()
ops.set_f32_gemm_mode(3)
for (bb, rr, tag) in ((None, None, "plain"), (b, None, "bias"), (b, r, "bias+res")):
    flag = ops.f16x2_begin("cuda")
    out = ops.gemm_nt(a, w, bb, rr, 0, cw=True).double()
    ref = a.double() @ w.double().T
    rs = a.double().abs() @ w.double().abs().T
    if bb is not None:
        ref += bb.double(); rs += bb.double().abs()
    if rr is not None:
        ref += rr.double(); rs += rr.double().abs()
    err = (out - ref).abs() / rs
    i = int(err.max(dim=1).values.argmax())
    print(tag, "max err/scale", err.max().item(), "worst row", i, "row mean|a|", a[i].abs().mean().item(), "row max|a|", a[i].abs().max().item(), "flag", int(flag.item()))
    # emulation of the split in f64
    h2, s = ops.h2_of(w)
    wh, wl = h2[0].double(), h2[1].double()
    v = a * 16.0
    ah = v.half(); al = (v - ah.float()).half()
    emu = (ah.double() @ wh.T + ah.double() @ wl.T + al.double() @ wh.T) / (16.0 * s)
    if bb is not None:
        emu += bb.double()
    if rr is not None:
        emu += rr.double()
    print("   emulation vs f64:", ((emu - ref).abs() / rs).max().item(), " GPU vs emulation:", ((out - emu).abs() / rs).max().item(),
          " GPU vs emulation on worst row:", ((out[i] - emu[i]).abs() / rs[i]).max().item())
    ops.set_f32_gemm_mode(0)
    o0 = ops.gemm_nt(a, w, bb, rr, 0).double()
    print("   native f32 vs f64:", ((o0 - ref).abs() / rs).max().item(), " row", int(((o0 - ref).abs() / rs).max(dim=1).values.argmax()))
    ops.set_f32_gemm_mode(3)
    # per row-scale bucket
    sc = a.abs().mean(1)
    for lo_, hi_ in ((0, 0.05), (0.05, 0.3), (0.3, 3), (3, 1000)):
        m = (sc >= lo_) & (sc < hi_)
        print(f"   rows with mean|a| in [{lo_},{hi_}): n={int(m.sum())} fp16x2 {err[m].max().item():.3e} native {(((o0 - ref).abs() / rs)[m]).max().item():.3e}")
ops.set_f32_gemm_mode(1)
