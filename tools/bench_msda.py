"""K1 micro-benchmark: achieved algorithmic GB/s of ovis_msda_forward_f32 at the encoder shape."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import MultiScaleDeformableAttention as MSDA

def inputs(B, sizes, dev, M=8, D=32, P=4, seed=0, spread=2.0):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(sizes, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum()); L = len(sizes)
    value = torch.randn(B, S, M, D, generator=g)
    ref = []
    for (H, W) in sizes:
        ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ref.append(torch.stack((xs.reshape(-1) / W, ys.reshape(-1) / H), -1))
    ref = torch.cat(ref, 0)
    off = torch.randn(B, S, M, L, P, 2, generator=g) * spread
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
    w = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    return [t.to(dev) for t in (value, shapes, lsi, loc, w)], S

if __name__ == "__main__":
    dev = "cuda:0"
    for name, B, sizes in (("720p_T5", 5, [(23, 40), (46, 80), (92, 160)]), ("480p_T1", 1, [(15, 27), (30, 54), (60, 108)]),
                           ("1080p_T5", 5, [(34, 60), (68, 120), (136, 240)])):
        args, S = inputs(B, sizes, dev)
        for _ in range(5):
            MSDA.ms_deform_attn_forward(*args, 128)
        torch.cuda.synchronize()
        n = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            MSDA.ms_deform_attn_forward(*args, 128)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        alg = 3200 * S * B
        print(json.dumps({"case": name, "ms": round(ms, 4), "alg_MB": alg / 1e6, "GBps": round(alg / ms / 1e6, 1)}))
