"""ResNet stem + pool at 5 x 736 x 1280: one fused launch against conv (gemm_f16cvt ConvA -> fp16) + fp16 pool."""
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
x = torch.randn(5, 736, 1280, 4, generator=g).cuda(); w = (torch.randn(64, 7, 8, 4, generator=g) / 12).half().cuda(); b = torch.randn(64, generator=g).cuda()
res = {}
for rnd in range(2):
    for xt in (1, 2, 4, 8, 20):
        _lib.call('ovis_stem_tiles', xt)
        res[xt] = timeit(lambda: ops.resnet_stem_pool(x, w, b))
    print(' '.join(f'xt={k}: {v:.1f}' for k, v in res.items()))
_lib.call('ovis_stem_tiles', 2)
t1 = res[2]
t2 = timeit(lambda: ops.maxpool3x3s2(ops.conv2d_nhwc_o16(x, w, 2, 3, b, ops.ACT_RELU)))
print(f"stem + pool fused {t1:.1f} us; conv + pool {t2:.1f} us")
