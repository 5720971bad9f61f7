import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openvis_amd import ops
from tools.bench_gemm import timeit
T, Q, H, W, Hp, Wp = 5, 100, 720, 1280, 736, 1280
g = torch.Generator().manual_seed(0)
frames = (torch.rand(T, 3, H, W, generator=g) * 255).to(torch.uint8).cuda()
masks = torch.randn(Q, T, Hp // 4, Wp // 4, generator=g).cuda()
for name, side in (("full-frame boxes", None), ("200px boxes", 200)):
    crops = []
    for t in range(T):
        for q in range(95):
            if side is None: crops.append([t, q, 0, 0, Wp - 1, Hp - 1])
            else:
                x0, y0 = (q * 37) % (W - side), (q * 53) % (H - side)
                crops.append([t, q, x0, y0, x0 + side - 1, y0 + side - 1])
    cr = torch.tensor(crops, dtype=torch.int32).cuda()
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    from openvis_amd import _lib
    for tile in (16, 8):
        _lib.call("ovis_crop_tile", 8 if tile == 8 else 0)
        ms = timeit(lambda: ops.clip_crop_patches(frames, masks, cr, Hp, Wp, 224, 16, mean, std, out_f16=True), n=10)
        print(name, len(crops), "crops", f"{tile}x{tile}-bin tiles", round(ms, 3), "ms")
    _lib.call("ovis_crop_tile", 0)
    # round 6: leader / follower passes (crops of a frame that share a box compute the frame half once) against the one fused pass
    for name2, sw in (("leader/follower passes (ships)", 0), ("one fused pass", 32)):
        _lib.call("ovis_crop_tile", sw)
        ms = timeit(lambda: ops.clip_crop_patches(frames, masks, cr, Hp, Wp, 224, 16, mean, std, out_f16=True), n=10)
        print(name, len(crops), "crops", name2, round(ms, 3), "ms")
    _lib.call("ovis_crop_tile", 0)
