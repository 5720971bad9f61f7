import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops
from tools.bench_gemm import timeit
sizes = [(23, 40), (46, 80), (92, 160)]
shapes = torch.tensor(sizes); lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum()); B = 5
g = torch.Generator().manual_seed(0)
value = torch.randn(B, S, 256, generator=g).cuda()
for scale in (0.25, 0.5, 1.0, 2.0, 4.0, 8.0):
    oa = torch.randn(B, S, 288, generator=g); oa[..., :192] *= scale; oa = oa.cuda()
    nbytes = 4.0 * B * S * (2 * 256 + 288)
    t0 = timeit(lambda: ops.msda_encoder_fused(value, oa, shapes.cuda(), lsi.cuda()), n=20)
    for R in (1, 2, 3, 4):
        ops.MSDA_TILE_RADIUS = R
        t1 = timeit(lambda: ops.msda_encoder_fused(value, oa, shapes.cuda(), lsi.cuda(), shapes_host=sizes), n=20)
        print(f"offset std {scale} px: direct {t0*1e3:.0f} us ({nbytes/t0/1e6:.0f} GB/s)   tiled R={R}: {t1*1e3:.0f} us ({nbytes/t1/1e6:.0f} GB/s)")
