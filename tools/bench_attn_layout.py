"""CLIP attention (500 crops x 12 heads x 197 tokens): the tower's interleaved layout (a head's K / V rows are 128-byte segments of 4 608-byte
token rows) against a head-major layout (every (crop, head)'s Q, K, V block contiguous: 197 x 128 B), emulated as B = 6 000, H = 1.
Question: would a QKV GEMM that writes head-major make the attention's staging (bandwidth-bound at 3.5 TB/s) stream faster?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, H, L, D = 500, 12, 197, 64
C = H * D
qkv = torch.randn(B * L, 3 * C, device="cuda").half()
hm = torch.randn(3, B * H, L, D, device="cuda").half()            # [q|k|v][crop * head][token][64]
for dbg in (0, 1, 2, 0):
    _lib.call("ovis_attention_f16_debug", dbg)
    t_il = timeit(lambda: ops.attention_f16(qkv, qkv[:, C:], qkv[:, 2 * C:], B, H, L, L, D, L * 3 * C, 3 * C, L * 3 * C, 3 * C, L * 3 * C, 3 * C))
    t_hm = timeit(lambda: ops.attention_f16(hm[0], hm[1], hm[2], B * H, 1, L, L, D, L * D, D, L * D, D, L * D, D))
    print(f"dbg={dbg} (0 full, 1 staging only, 2 no K/V loads): interleaved {t_il:.1f} us | head-major {t_hm:.1f} us", flush=True)
_lib.call("ovis_attention_f16_debug", 0)
