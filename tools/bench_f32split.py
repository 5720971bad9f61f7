"""The f32-class layers of the pixel decoder / FPN under each MODEL.F32_GEMM_SPLIT (ms per launch, same process, interleaved):
bf16x3 (6 bf16 products), bf16x2 (3 bf16 products, 16 bits per operand), fp16x2 (3 fp16 products, 22 bits per operand)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

NAMES = {1: "bf16x3", 2: "bf16x2", 3: "fp16x2"}


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(label, flops, fn):
    row = {"op": label}
    for rep in range(2):                                   # two rounds, interleaved: the second one is reported
        for m in (1, 2, 3):
            ops.set_f32_gemm_mode(m)
            if m == 3:
                ops.f16x2_begin("cuda")
            ms = timeit(fn)
            row[NAMES[m]] = {"ms": round(ms, 4), "TF": round(flops / ms / 1e9, 1)}
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    for (M, N, K, res) in [(96600, 256, 256, False), (96600, 256, 256, True), (96600, 1024, 256, False), (96600, 256, 1024, True),
                           (96600, 288, 256, False), (294400, 256, 256, False), (695520, 256, 256, False), (695520, 1024, 256, False)]:
        a = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev) if res else None
        run(f"gemm {M}x{N}x{K}{' +res' if res else ''}", 2.0 * M * N * K, lambda: ops.gemm_nt(a, w, b, r, 1 if not res else 0, cw=True))
        if res and N == 256:
            gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
            run(f"gemm+LN {M}x{N}x{K}", 2.0 * M * N * K, lambda: ops.gemm_nt_layernorm(a, w, b, r, gamma, beta))
        del a, w, b, r
    for (T, H, W, Cin, Cout) in [(5, 184, 320, 256, 256), (5, 92, 160, 256, 256), (36, 184, 320, 256, 256)]:
        x = torch.randn(T, H, W, Cin, generator=g).to(dev)
        gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
        xp = ops.groupnorm_nhwc(x, gamma, beta, pad=True)
        fl = 2.0 * T * H * W * Cout * 9 * Cin
        run(f"conv3x3 padded {T}x{H}x{W} {Cin}->{Cout}", fl, lambda: ops.conv3x3_padded(xp, w))
        run(f"conv3x3 gather {T}x{H}x{W} {Cin}->{Cout}", fl, lambda: ops.conv2d_nhwc(x, w, 1, 1, cw=True))
        del x, xp, w
