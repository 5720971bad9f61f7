"""final_masks_kernel: ten 5 x 720 x 1280 output masks from [100,5,184,320] logits (row-major and column-major)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops
from tools.bench_gemm import timeit
g = torch.Generator().manual_seed(0)
masks = torch.randn(100, 5, 184, 320, generator=g).cuda()
sel = torch.arange(10, dtype=torch.int32).cuda() * 7
for cm in (False, True):
    for (oh, ow) in ((720, 1280), (718, 1278), (1080, 1920)):
        t = timeit(lambda: ops.final_masks(masks, sel, 736, 1280, 720, 1280, oh, ow, column_major=cm), n=20)
        print(f"column_major={cm} out {oh}x{ow}: {t*1e3:.1f} us ({10*5*oh*ow/t/1e6:.0f} GB/s of output bytes)")
