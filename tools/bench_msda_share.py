"""K1 A/B in one process: msda_encoder_fused8_kernel (lane sharing) vs msda_encoder_fused_kernel, 5 x 720p frames, offsets of 2 / 4 / 8 px std."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openvis_amd import ops
from tools.bench_gemm import timeit
sizes = [(23, 40), (46, 80), (92, 160)]
shapes = torch.tensor(sizes); lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
S = int(shapes.prod(1).sum()); B = 5
g = torch.Generator().manual_seed(0)
value = torch.randn(B, S, 256, generator=g).cuda()
sh, ls = shapes.cuda(), lsi.cuda()
nbytes = 4.0 * B * S * (2 * 256 + 288)
for scale in (2.0, 4.0, 8.0):
    oa = torch.randn(B, S, 288, generator=g); oa[..., :192] *= scale; oa = oa.cuda()
    res = {}
    for rep in range(3):
        for share in (False, True):
            ops.msda_set_share(share)
            t = timeit(lambda: ops.msda_encoder_fused(value, oa, sh, ls), n=20)
            res[share] = min(res.get(share, 1e9), t)
    ops.msda_set_share(True)
    print(f"offset std {scale} px: every-lane kernel {res[False]*1e3:.1f} us ({nbytes/res[False]/1e6:.0f} GB/s) | lane-sharing kernel {res[True]*1e3:.1f} us ({nbytes/res[True]/1e6:.0f} GB/s)")
