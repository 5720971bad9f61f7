"""mask_bbox on the bench's shape (500 masks of 184 x 320 logits -> boxes at 736 x 1280): noise (random-init worst case) and blobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops, _lib
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
Q, T, h, w = 100, 5, 184, 320
noise = torch.randn(Q, T, h, w, generator=g).cuda()
yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
c = torch.rand(Q, T, 2, generator=g)
blobs = (3.0 - ((yy[None, None] - c[..., 0, None, None] * h) ** 2 + (xx[None, None] - c[..., 1, None, None] * w) ** 2) / (0.02 * h * w + 1)).contiguous().cuda()
for name, m in (("noise", noise), ("blobs", blobs)):
    ops.mask_bbox_set_cells(False); ref = ops.mask_bbox(m, 4 * h, 4 * w); t_ref = timeit(lambda: ops.mask_bbox(m, 4 * h, 4 * w)); ops.mask_bbox_set_cells(True)
    row = f"{name}: per-pixel kernel {t_ref:.0f} us;"
    for rows in (8, 24, 48, 92, 184):
        _lib.call("ovis_mask_bbox_set_rows", rows)
        same = torch.equal(ops.mask_bbox(m, 4 * h, 4 * w), ref)
        row += f"  rows {rows}: {timeit(lambda: ops.mask_bbox(m, 4 * h, 4 * w)):.0f} us{'' if same else ' MISMATCH'}"
    print(row, flush=True)
_lib.call("ovis_mask_bbox_set_rows", 24)
