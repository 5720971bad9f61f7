"""Per-launch times of the ResNet-50 backbone (A2) at the bench workload (5 x 736 x 1280), with each launch's algorithmic bytes:
   python tools/prof_backbone.py [fp16|fp32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openvis_amd import ops

prec = sys.argv[1] if len(sys.argv) > 1 else "mixed"
model, sd, _ = bench.build_model("cuda", precision=prec if prec in ("mixed", "fp32") else "mixed")
frames = bench.synth_frames(5, 720, 1280, 1000, "cuda")
images, _, _ = model.preprocess(frames)
bb = model.backbone
if os.environ.get("H16") == "0":
    bb.h16_storage = False            # the round-3 path: f32 storage everywhere
log = []
real_conv, real_gemm, real_pool = ops.conv2d_nhwc, ops.gemm_nt, ops.maxpool3x3s2


def timed(name, nbytes, flops, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = fn(); e1.record()
    log.append((name, nbytes, flops, e0, e1))
    return y


def conv(x, w, stride=1, pad=0, bias=None, residual=None, act=0, **kw):
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    nb = x.numel() * x.element_size() * (1 if stride == 1 or KH > 1 else 1.0 / stride ** 2) + N * OH * OW * Cout * 4 * (2 if residual is not None else 1)
    return timed(f"conv{KH}x{KW}s{stride} {H}x{W} {Cin}->{Cout}" + (" +res" if residual is not None else ""), nb, 2.0 * N * OH * OW * Cout * KH * KW * Cin,
                 lambda: real_conv(x, w, stride, pad, bias, residual, act, **kw))


def gemm(a, w, bias=None, residual=None, act=0, **kw):
    M, K = a.reshape(-1, a.shape[-1]).shape
    N = w.shape[0]
    nb = M * K * a.element_size() + M * N * 4 * (2 if residual is not None else 1)
    return timed(f"1x1 M={M} {K}->{N}" + (" +res" if residual is not None else ""), nb, 2.0 * M * N * K, lambda: real_gemm(a, w, bias, residual, act, **kw))


real_ch, real_x16, real_o16 = ops.conv_h16, ops.gemm_nt_x16, ops.conv2d_nhwc_o16


def conv_h16(x, w, k, stride=1, bias=None, residual=None, act=0, out_f16=True):
    T, H, W, Cin = x.shape
    Cout = w.shape[0]
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    nb = x.numel() * 2 + T * OH * OW * Cout * ((2 if out_f16 else 4) + (4 if residual is not None else 0))
    return timed(f"h16 conv{k}x{k}s{stride} {H}x{W} {Cin}->{Cout}" + (" +res" if residual is not None else "") + (" ->f16" if out_f16 else " ->f32"), nb,
                 2.0 * T * OH * OW * Cout * k * k * Cin, lambda: real_ch(x, w, k, stride, bias, residual, act, out_f16))


def x16(a, w, bias=None, residual=None, act=0, out_f16=False):
    M, K = a.reshape(-1, a.shape[-1]).shape
    N = w.shape[0]
    nb = M * K * a.element_size() + M * N * ((2 if out_f16 else 4) + (4 if residual is not None else 0))
    return timed(f"x16 1x1 M={M} {K}->{N} A{'16' if a.dtype == torch.float16 else '32'}" + (" +res" if residual is not None else "") + (" ->f16" if out_f16 else " ->f32"),
                 nb, 2.0 * M * N * K, lambda: real_x16(a, w, bias, residual, act, out_f16))


def o16(x, w, stride, pad, bias=None, act=0):
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    return timed(f"stem conv{KH}x{KW}s{stride} ->f16", x.numel() * 4 + N * OH * OW * Cout * 2, 2.0 * N * OH * OW * Cout * KH * KW * Cin, lambda: real_o16(x, w, stride, pad, bias, act))


ops.conv_h16, ops.gemm_nt_x16, ops.conv2d_nhwc_o16 = conv_h16, x16, o16
if len(sys.argv) > 2:
    from openvis_amd import _lib
    _lib.call("ovis_conv_h16_slots", int(sys.argv[2]))
if len(sys.argv) > 3:
    _lib.lib().ovis_conv_h16_bn(int(sys.argv[3]))
ops.conv2d_nhwc, ops.gemm_nt = conv, gemm
ops.maxpool3x3s2 = lambda x: timed(f"maxpool {x.shape[1]}x{x.shape[2]} C={x.shape[3]}", x.numel() * x.element_size() * 1.25, 0.0, lambda: real_pool(x))
for rep in range(3):
    log.clear()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); feats = bb(images); t1.record()
    torch.cuda.synchronize()
tot = 0.0
for name, nb, fl, e0, e1 in log:
    ms = e0.elapsed_time(e1)
    tot += ms
    print(f"{ms * 1e3:8.1f} us  {nb / 1e6:8.1f} MB  {nb / ms / 1e6:7.0f} GB/s  {fl / ms / 1e9:7.1f} TF  {name}")
print(f"sum of launches {tot:.3f} ms, backbone wall (events) {t0.elapsed_time(t1):.3f} ms, {len(log)} launches, total bytes {sum(l[1] for l in log) / 1e9:.2f} GB")
