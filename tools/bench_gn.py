"""Lab: GroupNorm (stats + apply) at the pixel decoder's shapes: time and a digest of the output (OVIS_LAB_LIB = a variant build)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import _lib
if os.environ.get("OVIS_LAB_LIB"):
    _lib.LIB_PATH = os.environ["OVIS_LAB_LIB"]
from openvis_amd import ops

g = torch.Generator().manual_seed(0)
for (T, H, W, C, pad, up, relu) in [(5, 184, 320, 256, True, True, False), (5, 184, 320, 256, False, False, True), (5, 92, 160, 256, False, False, False),
                                    (36, 184, 320, 256, True, True, False), (36, 184, 320, 256, False, False, True), (3, 61, 77, 128, True, True, True)]:
    x = torch.randn(T, H, W, C, generator=g).cuda()
    ga, be = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    u = torch.randn(T, (H + 1) // 2, (W + 1) // 2, C, generator=g).cuda() if up else None
    f = lambda: ops.groupnorm_nhwc(x, ga, be, relu=relu, up_add=u, pad=pad)
    y = f()
    dig = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"GroupNorm {T}x{H}x{W}x{C} pad={int(pad)} up={int(up)} relu={int(relu)}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  sha1 {dig}")
