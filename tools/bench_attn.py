"""Times ovis_attention_f16 on the CLIP ViT-B/16 (197 tokens) and ViT-L/14@336 (577 tokens) shapes of the bench workloads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops

from openvis_amd import _lib
for (B, H, L), dbg in [((500, 12, 197), 0), ((500, 12, 197), 0), ((500, 12, 197), 1), ((500, 12, 197), 2), ((180, 16, 577), 0)]:
    _lib.call("ovis_attention_f16_debug", dbg)
    D, C = 64, H * 64
    qkv = torch.randn(B * L, 3 * C, device="cuda").half()
    f = lambda: ops.attention_f16(qkv, qkv[:, C:], qkv[:, 2 * C:], B, H, L, L, D, L * 3 * C, 3 * C, L * 3 * C, 3 * C, L * 3 * C, 3 * C)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = (B * L * 3 * C * 2 + B * L * C * 2) / 1e9
    _lib.call("ovis_attention_f16_debug", 0)
    print(f"dbg={dbg} (0 full, 1 staging only, 2 no K/V loads) B={B} H={H} L={L}: {ms*1e3:.1f} us  {4.0*B*H*L*L*D/ms/1e9:.0f} TF  {gb/ms:.2f} TB/s")
