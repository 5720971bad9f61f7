"""conv_h16_kernel on the four 3x3 shapes of ResNet-50 (5 x 736 x 1280): whole kernel, DMA only (dbg 2), compute only (dbg 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator().manual_seed(0)
for (H, W, C) in [(184, 320, 64), (92, 160, 128), (46, 80, 256), (23, 40, 512)]:
    x = torch.randn(5, H, W, C, generator=g).relu().half().cuda()
    w = (torch.randn(C, 3, 3, C, generator=g) / (9 * C) ** 0.5).half().cuda()
    b = torch.randn(C, generator=g).cuda()
    row = f"{H}x{W} C={C}:"
    for slots in (2, 3):
        _lib.call("ovis_conv_h16_slots", slots)
        for dbg, name in ((0, "all"), (2, "dma"), (1, "mfma")):
            _lib.lib().ovis_conv_h16_debug(dbg)
            row += f"  s{slots}/{name} {timeit(lambda: ops.conv_h16(x, w, 3, 1, b, None, 1, True)):6.1f}"
    _lib.lib().ovis_conv_h16_debug(0)
    print(row, flush=True)
