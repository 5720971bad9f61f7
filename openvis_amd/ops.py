"""Thin torch-tensor wrappers over the C-ABI entry points of libopenvis_hip.so.

torch is used for device memory, streams and shape bookkeeping only; all arithmetic happens in
the hand-written gfx950 kernels.  Every function requires CUDA(HIP) tensors and raises otherwise.
"""
import ctypes

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_QUICKGELU = 0, 1, 2


def _ll(v):
    return ctypes.c_longlong(int(v))


def _chk(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise _lib.OvisError("openvis_amd ops need contiguous HIP device tensors (no CPU fallback)")


def gemm_nt(a, w, bias=None, residual=None, act=ACT_NONE, out=None):
    """out[m,n] = act(sum_k a[m,k] w[n,k] + bias[n] + residual[m,n]); a [...,K] -> out [...,N]."""
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    _chk(a2, w, bias, residual)
    M = a2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    _lib.call("ovis_gemm_nt_f32", a2, _ll(K), w, _ll(w.stride(0)), out, _ll(N), M, N, K, bias, r2, _ll(N), act,
              _lib.stream_ptr())
    return out.view(*a.shape[:-1], N)


def conv2d_nhwc(x, w, stride=1, pad=0, bias=None, residual=None, act=ACT_NONE):
    """x [N,H,W,Cin], w [Cout,KH,KW,Cin] -> [N,OH,OW,Cout]."""
    _chk(x, w, bias, residual)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device)
    _lib.call("ovis_conv2d_nhwc_f32", x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, bias, residual, act,
              _lib.stream_ptr())
    return y
