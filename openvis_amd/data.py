"""Input side of the eval path: test-time resize of decoded frames on the GPU.

The reference's test mapper resizes every decoded frame with detectron2's ResizeShortestEdge(INPUT.MIN_SIZE_TEST,
INPUT.MAX_SIZE_TEST) -> PIL `Image.resize(BILINEAR)` on the CPU (openvis/data/augmentation.py:368-373;
ytvis_dataset_mapper.py:298-313) and uploads the result.  Here the decoded uint8 HWC frame is uploaded as is and
resampled by csrc/resize.hip with Pillow's exact arithmetic; only the small coefficient tables are computed on the host."""
import functools
import math

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2


def shortest_edge_size(h, w, min_size, max_size=1333):
    """detectron2 ResizeShortestEdge.get_output_shape (v0.6 transforms/augmentation_impl.py)."""
    if min_size == 0:
        return h, w
    scale = min_size * 1.0 / min(h, w)
    newh, neww = (min_size, scale * w) if h < w else (scale * h, min_size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


@functools.lru_cache(maxsize=64)
def pil_bilinear_coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs (triangle filter, support 1) + normalize_coeffs_8bpc -> (bounds int32 [out,2],
    coefficients int32 [out,ksize], ksize).  Python floats are C doubles, so the tables are bit-identical."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        kk[xx, :xmax] = [int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_frame(frame_hwc_u8, out_hw):
    """uint8 [H,W,3] device tensor (decoded frame) -> uint8 [3,OH,OW] on the device, identical to
    np.asarray(Image.fromarray(frame).resize((OW, OH), Image.BILINEAR)).transpose(2, 0, 1)."""
    if not (frame_hwc_u8.is_cuda and frame_hwc_u8.dtype == torch.uint8 and frame_hwc_u8.is_contiguous()):
        raise _lib.OvisError("resize_frame needs a contiguous uint8 HWC device tensor")
    H, W, _ = frame_hwc_u8.shape
    OH, OW = out_hw
    dev = frame_hwc_u8.device
    xb, xk, xks = pil_bilinear_coeffs(W, OW)
    yb, yk, yks = pil_bilinear_coeffs(H, OH)
    t = lambda a: torch.from_numpy(a).to(dev)
    tmp = torch.empty((H, OW, 3), dtype=torch.uint8, device=dev)
    dst = torch.empty((3, OH, OW), dtype=torch.uint8, device=dev)
    _lib.call("ovis_pil_resize_u8_hwc_to_chw", frame_hwc_u8, H, W, tmp, dst, OH, OW, t(xb), t(xk), xks, t(yb), t(yk), yks,
              _lib.stream_ptr())
    return dst


def resize_and_preprocess(frames_hwc_u8, min_size, max_size=1333, size_divisibility=32, pixel_mean=(123.675, 116.28, 103.53),
                          pixel_std=(58.395, 57.12, 57.375), device="cuda"):
    """SURVEY.md 8f-2, fused: list of decoded uint8 [H,W,3] frames of ONE video -> (frames uint8 [T,3,OH,OW] -- one device tensor, what the model's
    `image` list slices --, images f32 [T,Hp,Wp,4] = the model's A1 output for them, original (H, W)).  The vertical pass of the resize writes both
    (csrc/resize.hip: resize_vertical_preprocess_kernel); hand `images` to the forward as batched_inputs[0]["images_nhwc4"] and A1 is skipped.
    Bit-identical to resize_frame() + VideoMaskFormer.preprocess()."""
    fs = [torch.as_tensor(np.ascontiguousarray(f)) if not torch.is_tensor(f) else f for f in frames_hwc_u8]
    H, W, _ = fs[0].shape
    OH, OW = shortest_edge_size(H, W, min_size, max_size)
    d = size_divisibility
    Hp, Wp = ((OH + d - 1) // d * d, (OW + d - 1) // d * d) if d > 1 else (OH, OW)
    xb, xk, xks = pil_bilinear_coeffs(W, OW)
    yb, yk, yks = pil_bilinear_coeffs(H, OH)
    t = lambda a: torch.from_numpy(a).to(device)
    xb, xk, yb, yk = t(xb), t(xk), t(yb), t(yk)
    T = len(fs)
    frames = torch.empty((T, 3, OH, OW), dtype=torch.uint8, device=device)
    images = torch.empty((T, Hp, Wp, 4), dtype=torch.float32, device=device)
    tmp = torch.empty((H, OW, 3), dtype=torch.uint8, device=device)
    from .ops import _f3
    for i, f in enumerate(fs):
        if tuple(f.shape) != (H, W, 3) or f.dtype != torch.uint8:
            raise _lib.OvisError("resize_and_preprocess: all frames of a video are uint8 [H,W,3] of one size")
        src = f.to(device, non_blocking=True).contiguous()
        _lib.call("ovis_pil_resize_preprocess_u8", src, H, W, tmp, frames[i], images[i], OH, OW, Hp, Wp, xb, xk, xks, yb, yk, yks,
                  _f3(pixel_mean), _f3(pixel_std), _lib.stream_ptr())
    return frames, images, (H, W)


def load_and_resize(frames_hwc_u8, min_size, max_size=1333, device="cuda"):
    """list of decoded uint8 [H,W,3] arrays/tensors -> (list of uint8 [3,OH,OW] device tensors, original (H, W))."""
    out = []
    for f in frames_hwc_u8:
        f = torch.as_tensor(np.ascontiguousarray(f)) if not torch.is_tensor(f) else f
        H, W, _ = f.shape
        out.append(resize_frame(f.to(device, non_blocking=True).contiguous(), shortest_edge_size(H, W, min_size, max_size)))
    return out, (H, W)
