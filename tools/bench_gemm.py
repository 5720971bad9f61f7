"""f32 MFMA GEMM / conv micro-benchmark (TFLOP/s) at the path's shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

if __name__ == "__main__":
    dev = "cuda:0"
    for (M, N, K) in [(96600, 256, 256), (96600, 1024, 256), (96600, 256, 1024), (96600, 288, 256), (4096, 4096, 4096),
                      (294400, 256, 256), (98500, 2304, 768), (98500, 3072, 768), (100, 256, 256)]:
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        ops.set_f32_gemm_mode(0)
        ms0 = timeit(lambda: ops.gemm_nt(a, w, b, None, 1))
        ops.set_f32_gemm_mode(1)
        ms = timeit(lambda: ops.gemm_nt(a, w, b, None, 1))
        ms_t = timeit(lambda: torch.relu(torch.nn.functional.linear(a, w, b)))
        print(json.dumps({"gemm": [M, N, K], "ms": round(ms, 4), "TF": round(2 * M * N * K / ms / 1e9, 1), "native_f32_TF": round(2 * M * N * K / ms0 / 1e9, 1), "torch_ms": round(ms_t, 4),
                          "torch_TF": round(2 * M * N * K / ms_t / 1e9, 1)}))
    for (N, H, W, Cin, Cout, k, s, p) in [(5, 184, 320, 256, 256, 3, 1, 1), (5, 184, 320, 64, 64, 3, 1, 1), (5, 736, 1280, 4, 64, 7, 2, 3),
                                          (5, 92, 160, 128, 128, 3, 1, 1), (5, 46, 80, 256, 256, 3, 1, 1), (5, 23, 40, 512, 512, 3, 1, 1)]:
        x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev)
        ops.set_f32_gemm_mode(0)
        ms0 = timeit(lambda: ops.conv2d_nhwc(x, w, s, p, None, None, 1))
        ops.set_f32_gemm_mode(1)
        ms = timeit(lambda: ops.conv2d_nhwc(x, w, s, p, None, None, 1))
        OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
        fl = 2 * N * OH * OW * Cout * Cin * k * k
        xc = x.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last); wc = w.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        ms_t = timeit(lambda: torch.relu(torch.nn.functional.conv2d(xc, wc, None, s, p)))
        print(json.dumps({"conv": [N, H, W, Cin, Cout, k, s], "ms": round(ms, 4), "TF": round(fl / ms / 1e9, 1), "native_f32_TF": round(fl / ms0 / 1e9, 1), "torch_ms": round(ms_t, 4),
                          "torch_TF": round(fl / ms_t / 1e9, 1)}))
