"""CLIP cosine-logit error with IDENTICAL crops on both sides (isolates A10/A12 from mask flips): synthetic low-res mask logits
are given to the HIP ClipAdapter (boxes -> crops -> ViT-B/16 -> x100 logits) and, up-sampled x4 like openvis.py:87-96, to the
oracle (clip_crops -> clip_encode_image).  Prints the distribution of |delta cos| per operand precision of the tower.

  python tools/exp_logit_bound.py [n_queries]  > gpurun_out/logit_bound.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

import bench
from openvis_amd import weights
from openvis_amd.modeling.clip_adapter.adapter import ClipAdapter, _CLIP_ARCH
from oracle import torch_ref as TR


def synthetic_masks(Q, T, h, w, seed=5):
    """smooth blobs with steep edges (logit +-12 a few low-res pixels apart): every up-sampled pixel is far from the 0.5
    threshold, so both sides derive the same boxes."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    m = torch.empty(Q, T, h, w)
    for q in range(Q):
        cy, cx = torch.rand(2, generator=g).tolist()
        ry, rx = (0.05 + 0.3 * torch.rand(2, generator=g)).tolist()
        for t in range(T):
            d = ((yy - (cy + 0.01 * t) * h) / (ry * h)) ** 2 + ((xx - (cx - 0.01 * t) * w) / (rx * w)) ** 2
            m[q, t] = (1.0 - d) * 12.0
    return m.clamp(-12, 12)


def run(Q=40, T=2, K=482, device="cuda"):
    arch = _CLIP_ARCH["ViT-B/16"]
    sd = weights.random_init(weights.clip_visual_spec(**arch), seed=42)
    frames = bench.synth_frames(T, 720, 1280, 3, "cpu")
    Hp, Wp = 736, 1280
    masks = synthetic_masks(Q, T, Hp // 4, Wp // 4)
    text = bench.synth_text(K, 512)
    names = [f"class_{i}" for i in range(K)]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    t0 = time.time()
    with torch.no_grad():
        up = F.interpolate(masks, size=(Hp, Wp), mode="bilinear", align_corners=False)
        part = up.sigmoid().transpose(0, 1).contiguous()                       # [T,Q,Hp,Wp]
        regions, valid, boxes = TR.clip_crops(frames, part)
        feat = TR.clip_encode_image(regions, sd)
        ref = feat @ text.T                                                    # cosines [M,K]
    t_ref = time.time() - t0
    res = {}
    for prec in ("fp16 + fp16 stream", "fp16", "fp32"):      # first = the bench policy (MODEL.CLIP_ADAPTER.RESIDUAL_STREAM fp16)
        ad = ClipAdapter("ViT-B/16", precision=prec.split()[0]).load_state_dict(sd, "clip_adapter.", device)
        ad.visual.stream16 = prec.endswith("stream")
        ad.set_text_features(names, text)
        logits, v, crops = ad(frames.to(device), names, masks.to(device), (Hp, Wp))
        assert (v == valid.numpy()).all() and logits.shape == ref.shape
        d = (logits.cpu() / 100.0 - ref).abs()
        res[prec] = d
    return res, ref, t_ref


if __name__ == "__main__":
    Q = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    res, ref, t_ref = run(Q)
    print(f"# {ref.shape[0]} crops x {ref.shape[1]} classes, oracle {t_ref:.1f} s; |delta cos| of the HIP tower vs the f32 oracle on identical crops")
    for prec, d in res.items():
        per_crop = d.max(dim=1).values.numpy()
        print(f"tower operands {prec}: max {d.max().item():.3e}  p99.9 {np.quantile(d.numpy(), 0.999):.3e}  median {d.median().item():.3e}; "
              f"per-crop max: median {np.median(per_crop):.3e} p90 {np.quantile(per_crop, 0.9):.3e}; crops above 1e-3: {(per_crop > 1e-3).sum()}")
