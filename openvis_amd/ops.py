"""Thin torch-tensor wrappers over the C-ABI entry points of libopenvis_hip.so.

torch is used for device memory, streams and shape bookkeeping only; all arithmetic happens in
the hand-written gfx950 kernels.  Every function requires CUDA(HIP) tensors and raises otherwise.
"""
import ctypes

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_QUICKGELU, ACT_GELU = 0, 1, 2, 3

_EXT = None


def _mi():
    """torch.ops.ovis_mi: the dispatcher ops of the hot stages (csrc/torch_ext/hot_ops.cpp + msda_module.cpp, registered when the
    compiled extension `MultiScaleDeformableAttention` is imported).  A missing build fails loudly, with the build hint."""
    global _EXT
    if _EXT is None:
        from .modeling.pixel_decoder.ops.functions import ms_deform_attn_func      # imports the extension (or raises the hint)
        if not hasattr(torch.ops.ovis_mi, "gemm_nt_f16"):
            raise _lib.OvisError("the compiled extension lacks the ovis_mi hot ops: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _EXT = torch.ops.ovis_mi
    return _EXT

# bench.py sets this to a list to time every MFMA GEMM launch with HIP events on the launch stream:
# entries are (kernel name, algorithmic flops, start event, end event).
PROFILE = None


class _Mode(__import__("threading").local):
    """the library keeps the f32-GEMM split per host thread (thread_local in csrc/gemm_f32.hip): so does this mirror of it"""
    v = 1
    flag = None          # fp16x2: the device int the kernels of this thread raise when an operand left the fp16 range (f16x2_begin)
    a_scale = 16.0       # fp16x2: power of two the activations are multiplied by while they are split


_MODE = _Mode()


def _gemm_variant(M, N, loader, K=4, batch=1, h2=False):
    big = ((M + 127) // 128) * ((N + 127) // 128) * batch >= 256       # mirrors launch_gemm() in csrc/gemm_f32.hip
    if big and _MODE.v >= 1 and K % 4 == 0:
        return f"gemm_f32x3_kernel<128,128,{loader},FH>" if h2 else f"gemm_f32x3_kernel<128,128,{loader}>"
    return f"gemm_f32_kernel<{'128,128' if big else '64,64'},{loader}>"


class _Prof:
    def __init__(self, name, flops, unit="flop"):
        self.name, self.flops = (name if unit == "flop" else "hbm:" + name), flops   # "hbm:" entries carry BYTES

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1))


def _ll(v):
    return ctypes.c_longlong(int(v))


def _chk(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise _lib.OvisError("openvis_amd ops need contiguous HIP device tensors (no CPU fallback)")


def to_device_async(arr, device):
    """small host array (numpy) -> device tensor WITHOUT blocking the host: through a pinned staging buffer (torch's caching host
    allocator keeps it alive until the copy has run).  `torch.from_numpy(a).to(device)` copies from pageable memory, which blocks the
    host until everything queued on the stream before it has finished -- at the end of the CLIP stage that is 20 ms during which the
    host could have queued the next clip."""
    t = torch.from_numpy(arr) if not torch.is_tensor(arr) else arr
    if not torch.device(device).type == "cuda":
        return t.to(device)
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    return p.to(device, non_blocking=True)


def f32_gemm_mode():
    return _MODE.v


def set_f32_gemm_mode(mode):
    """0: native f32 MFMA for every f32 GEMM/conv; 1 (library default): large problems use the exact bf16x3 split (gemm_f32x3.h);
    2: the same kernel with the two leading planes only (bf16x2, three products); 3: "fp16x2" -- constant-weight layers (cw=True) run the
    three products hi hi + hi lo + lo hi of the fp16 split (11 + 11 significand bits: f32 grade at the MFMA cost of bf16x2; the *_h2 entry
    points of include/openvis_hip.h), everything else stays on bf16x3.  Per HOST THREAD (MODEL.F32_GEMM_SPLIT): two models with different
    splits on different ClipPipeline threads do not see each other's setting."""
    _lib.call("ovis_set_f32_gemm_mode", int(mode))
    _MODE.v = int(mode)


def f16x2_begin(device, a_scale=None):
    """fp16x2: a fresh zeroed range flag for the work this thread queues from now on (one per forward: the previous forward's flag may still
    be waiting for its device -> host copy) and the activation scale.  Returns the flag tensor (int32 [1])."""
    if a_scale is not None:
        _MODE.a_scale = float(a_scale)
    _MODE.flag = torch.zeros((1,), dtype=torch.int32, device=device)
    _lib.call("ovis_set_f16x2", ctypes.c_float(_MODE.a_scale), _MODE.flag)
    return _MODE.flag


def f16x2_flag():
    """the range flag of this thread's current forward (None outside the fp16x2 policy)"""
    return _MODE.flag if _MODE.v == 3 else None


import weakref

_W3_CACHE = {}      # id(constant f32 weight tensor) -> (weakref to it, its three bf16 planes); split once per tensor


def _w3_kernel_name(a, lda, w3, ldb, plane, out, ldc, M, N, K, bias, residual, ldr, act):
    """The kernel ovis_gemm_nt_f32_w3 / the 1x1 case of ovis_conv2d_nhwc_f32_w3 would launch ("" = gemm_f32x3_kernel)."""
    fn = _lib.lib().ovis_gemm_nt_f32_w3_kernel
    fn.restype = ctypes.c_char_p
    return fn(_lib._conv(a), _ll(lda), _lib._conv(w3), _ll(ldb), _ll(plane), _lib._conv(out), _ll(ldc), M, N, K, _lib._conv(bias),
              _lib._conv(residual), _ll(ldr), act).decode()


def w3_of(w):
    """The exact 3-way bf16 split of a CONSTANT weight tensor (csrc/gemm_f32x3.h), cached per (base tensor, view geometry): a
    `w.view(...)` or a row slice `w[C:]` made afresh at every call hits the same entry; entries die with the base tensor."""
    base = w._base if w._base is not None else w
    key = (id(base), w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    hit = _W3_CACHE.get(key)
    if hit is not None and hit[0]() is base:
        return hit[1]
    _chk(w)
    p = torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    _lib.call("ovis_split_f32_to_bf16x3", w, p, _ll(w.numel()), _lib.stream_ptr())
    _W3_CACHE[key] = (weakref.ref(base, lambda _r, k=key: _W3_CACHE.pop(k, None)), p)
    return p


_H2_CACHE = {}      # like _W3_CACHE: (two fp16 planes of w * scale, scale)


def h2_of(w):
    """The fp16x2 operand of a CONSTANT weight tensor: fp16 [2, *w.shape] = (hi, lo) of w * scale with scale = 2^k such that
    max |w| * scale lies in [2^14, 2^15) -- the top of the fp16 range, so that lo keeps its 11 bits far below every weight that matters
    (absolute floor 2^-25 against 2^14).  Cached like w3_of; the scale costs one host read-back per weight tensor, at its first use."""
    base = w._base if w._base is not None else w
    key = (id(base), w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    hit = _H2_CACHE.get(key)
    if hit is not None and hit[0]() is base:
        return hit[1], hit[2]
    _chk(w)
    amax = float(w.abs().max().item())
    scale = 1.0 if not (amax > 0.0 and amax < float("inf")) else 2.0 ** (14 - __import__("math").frexp(amax)[1] + 1)
    p = torch.empty((2,) + tuple(w.shape), dtype=torch.float16, device=w.device)
    _lib.call("ovis_split_f32_to_f16x2", w, p, _ll(w.numel()), ctypes.c_float(scale), _lib.stream_ptr())
    _H2_CACHE[key] = (weakref.ref(base, lambda _r, k=key: _H2_CACHE.pop(k, None)), p, scale)
    return p, scale


def _h2_kernel_name(a, lda, h2, ldb, plane, out, ldc, M, N, K, bias, residual, ldr, act):
    fn = _lib.lib().ovis_gemm_nt_f32_h2_kernel
    fn.restype = ctypes.c_char_p
    return fn(_lib._conv(a), _ll(lda), _lib._conv(h2), _ll(ldb), _ll(plane), _lib._conv(out), _ll(ldc), M, N, K, _lib._conv(bias),
              _lib._conv(residual), _ll(ldr), act).decode()


def gemm_nt(a, w, bias=None, residual=None, act=ACT_NONE, out=None, w16=None, cw=False):
    """out[m,n] = act(sum_k a[m,k] w[n,k] + bias[n] + residual[m,n]); a [...,K] -> out [...,N].
    w16 (an fp16 copy of w) selects the autocast arithmetic: operands rounded to fp16, f32 accumulation.
    cw=True: w is a constant weight (not an activation) -> its bf16x3 planes are split once and cached (same results)."""
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    _chk(a2, w, bias, residual, w16)
    M = a2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    if w16 is not None and K % 8 == 0:
        big = ((M + 127) // 128) * ((N + 127) // 128) >= 1024 and N > 64         # mirrors launch() in csrc/gemm_f16cvt.hip
        with _Prof(f"gemm_f16cvt_kernel<{'128,128' if big else '64,64'},DenseA>", 2.0 * M * N * K):
            _lib.call("ovis_gemm_nt_f32a_f16w", a2, _ll(K), w16, _ll(K), out, _ll(N), M, N, K, bias, r2, _ll(N), act,
                      _lib.stream_ptr())
        return out.view(*a.shape[:-1], N)
    use_w3 = cw and w.is_contiguous() and K % 8 == 0 and _MODE.v >= 1 and ((M + 127) // 128) * ((N + 127) // 128) >= 256
    if use_w3 and _MODE.v == 3:                              # fp16x2: two fp16 planes of w * scale, three fp16 MFMA products
        h2, ws = h2_of(w)
        label = ""
        if PROFILE is not None:
            label = _h2_kernel_name(a2, K, h2, K, w.numel(), out, N, M, N, K, bias, r2, N, act) or _gemm_variant(M, N, "DenseA", K, h2=True)
        with _Prof(label, 2.0 * M * N * K):
            _lib.call("ovis_gemm_nt_f32_h2", a2, _ll(K), w, _ll(K), h2, _ll(w.numel()), ctypes.c_float(ws), out, _ll(N), M, N, K, bias, r2,
                      _ll(N), act, _lib.stream_ptr())
        return out.view(*a.shape[:-1], N)
    label = _gemm_variant(M, N, "DenseA", K) if PROFILE is not None else ""
    if PROFILE is not None and use_w3 and _MODE.v == 2:      # bf16x2: the ping-pong kernel's f32-A mode takes the eligible shapes
        label = _w3_kernel_name(a2, K, w3_of(w), K, w.numel(), out, N, M, N, K, bias, r2, N, act) or label
    with _Prof(label, 2.0 * M * N * K):
        if use_w3:
            _lib.call("ovis_gemm_nt_f32_w3", a2, _ll(K), w, _ll(K), w3_of(w), _ll(w.numel()), out, _ll(N), M, N, K, bias, r2,
                      _ll(N), act, _lib.stream_ptr())
        else:
            _lib.call("ovis_gemm_nt_f32", a2, _ll(K), w, _ll(w.stride(0)), out, _ll(N), M, N, K, bias, r2, _ll(N), act,
                      _lib.stream_ptr())
    return out.view(*a.shape[:-1], N)


def gemm_nt_dual(a, w, bias, r, col0):
    """One GEMM, two outputs (fp16x2 only): c1 = a w[:col0]^T + bias[:col0], c2 = a w[col0:]^T + bias[col0:] + r[m % r.shape[0]] -- two
    projections of the same rows whose inputs differ by a row-periodic term (ovis_gemm_nt_f32_h2_dual).  Returns (c1, c2), or None when
    MODEL.F32_GEMM_SPLIT is not fp16x2 or the kernel does not take the shape: the caller then runs the two GEMMs."""
    if _MODE.v != 3 or not w.is_contiguous():
        return None
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    _chk(a2, w, bias, r)
    M = a2.shape[0]
    h2, ws = h2_of(w)
    c1 = torch.empty((M, col0), dtype=torch.float32, device=a.device)
    c2 = torch.empty((M, N - col0), dtype=torch.float32, device=a.device)
    args = (_lib._conv(a2), _ll(K), _lib._conv(h2), _ll(K), _ll(w.numel()))
    tail = (_lib._conv(c1), _ll(col0), _lib._conv(c2), _ll(N - col0), M, N, K, _lib._conv(bias), _lib._conv(r), _ll(r.shape[-1]), r.shape[0], col0)
    if not _lib.lib().ovis_gemm_nt_f32_h2_dual_eligible(*args, *tail):
        return None
    with _Prof("gemm_f16_pp_kernel<0,0,true,false,true,false,DUAL,FH>", 2.0 * M * N * K):
        _lib.call("ovis_gemm_nt_f32_h2_dual", a2, _ll(K), h2, _ll(K), _ll(w.numel()), ctypes.c_float(ws), c1, _ll(col0), c2, _ll(N - col0), M, N, K,
                  bias, r, _ll(r.shape[-1]), r.shape[0], col0, _lib.stream_ptr())
    return c1.view(*a.shape[:-1], col0), c2.view(*a.shape[:-1], N - col0)


def gemm_nt_layernorm(a, w, bias, residual, gamma, beta, eps=1e-5):
    """LayerNorm(a w^T + bias + residual) over the last dimension (post-norm of the pixel decoder's encoder layers, msdeformattn.py:139-146).
    Where the bf16x2 ping-pong kernel takes the GEMM and N == 256, the LayerNorm runs in that kernel's epilogue (ovis_gemm_nt_f32_w3_ln:
    the [M, N] tensor makes one trip to memory instead of three); otherwise gemm_nt + layernorm -- the same arithmetic up to the order
    of the f32 row sums."""
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    r2 = residual.reshape(-1, N)
    _chk(a2, w, bias, r2, gamma, beta)
    M = a2.shape[0]
    if w.is_contiguous() and N == 256 and _MODE.v == 3 and K % 8 == 0 and ((M + 127) // 128) * ((N + 127) // 128) >= 256:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        h2, ws = h2_of(w)
        if _lib.lib().ovis_gemm_nt_f32_h2_ln_eligible(_lib._conv(a2), _ll(K), _lib._conv(h2), _ll(K), _ll(w.numel()), _lib._conv(out), _ll(N), M, N, K,
                                                      _lib._conv(bias), _lib._conv(r2), _ll(N)):
            with _Prof("gemm_f16_pp_kernel<0,0,true,false,true,false,FH>", 2.0 * M * N * K):
                _lib.call("ovis_gemm_nt_f32_h2_ln", a2, _ll(K), h2, _ll(K), _ll(w.numel()), ctypes.c_float(ws), out, _ll(N), M, N, K, bias, r2, _ll(N),
                          gamma, beta, ctypes.c_float(eps), _lib.stream_ptr())
            return out.view(*a.shape[:-1], N)
    if w.is_contiguous() and N == 256 and _MODE.v == 2 and K % 8 == 0 and ((M + 127) // 128) * ((N + 127) // 128) >= 256:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        w3 = w3_of(w)
        if _lib.lib().ovis_gemm_nt_f32_w3_ln_eligible(_lib._conv(a2), _ll(K), _lib._conv(w3), _ll(K), _ll(w.numel()), _lib._conv(out), _ll(N), M, N, K,
                                                      _lib._conv(bias), _lib._conv(r2), _ll(N)):
            with _Prof("gemm_f16_pp_kernel<0,0,true,false,true,false>", 2.0 * M * N * K):
                _lib.call("ovis_gemm_nt_f32_w3_ln", a2, _ll(K), w3, _ll(K), _ll(w.numel()), out, _ll(N), M, N, K, bias, r2, _ll(N), gamma, beta,
                          ctypes.c_float(eps), _lib.stream_ptr())
            return out.view(*a.shape[:-1], N)
    return layernorm(gemm_nt(a, w, bias, residual, cw=True), gamma, beta, eps=eps)


def gemm_nt_f16(a, w, bias=None, residual=None, act=ACT_NONE, out_f16=False):
    """fp16 a [...,K] x fp16 w [N,K] -> f32 (or fp16) [...,N]; f32 accumulation / bias / residual."""
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    _chk(a2, w, bias, residual)
    if a2.dtype != torch.float16 or w.dtype != torch.float16:
        raise _lib.OvisError("gemm_nt_f16 needs fp16 operands")
    M = a2.shape[0]
    r2 = residual.reshape(-1, N) if residual is not None else None
    if r2 is not None and r2.dtype == torch.float16:
        # the tower's fp16 residual stream: C (fp16) = A B^T + bias + R (fp16), accumulated in f32
        if act != ACT_NONE:
            raise _lib.OvisError("gemm_nt_f16: an fp16 residual goes with act = none (out-proj / c_proj)")
        if w.is_contiguous() and _lib.lib().ovis_gemm_nt_f16_res16_eligible(_lib._conv(r2), _lib._conv(r2), _ll(K), _ll(K), _ll(N), _ll(N), M, N, K,
                                                                              _lib._conv(bias)):
            with _Prof("gemm_f16_pp_kernel<1,0,true,false,false,true>", 2.0 * M * N * K):
                out = _mi().gemm_nt_f16(a2, w, bias, r2, ACT_NONE, True)
            return out.view(*a.shape[:-1], N)
        # small problems (few crops): the same arithmetic through the f32-residual kernels and one rounding at the end
        o32 = gemm_nt_f16(a2, w, bias, cast_f16_to_f32_rows(r2, N), act, out_f16=False)
        return cast_f16(o32).view(*a.shape[:-1], N)
    kname = ""
    if PROFILE is not None:                     # the kernel the library picks (name as rocprofv3 reports it)
        fn = _lib.lib().ovis_gemm_nt_f16_kernel
        fn.restype = ctypes.c_char_p
        probe = torch.empty((M, N), dtype=torch.float16 if out_f16 else torch.float32, device=a.device)
        kname = fn(_lib._conv(probe), _ll(K), _ll(w.stride(0)), _ll(N), M, N, K, _lib._conv(bias), _lib._conv(r2), _ll(N), act,
                   int(out_f16)).decode()
    with _Prof(kname, 2.0 * M * N * K):
        if w.is_contiguous():
            out = _mi().gemm_nt_f16(a2, w, bias, r2, act, bool(out_f16))
        else:                                   # row-strided weight view (e.g. the K / V rows of in_proj_weight): C ABI with ldb
            out = torch.empty((M, N), dtype=torch.float16 if out_f16 else torch.float32, device=a.device)
            _lib.call("ovis_gemm_nt_f16", a2, _ll(K), w, _ll(w.stride(0)), out, _ll(N), M, N, K, bias, r2, _ll(N), act,
                      int(out_f16), _lib.stream_ptr())
    return out.view(*a.shape[:-1], N)


def fold_layernorm(w, bias, gamma, beta):
    """LayerNorm folded into the Linear that consumes it (include/openvis_hip.h, ovis_gemm_nt_f16_ln): f32 w [N,K], bias [N] or None,
    gamma / beta [K] -> (wg fp16 [N,K] = fp16(gamma * w), s f32 [N] = row sums of wg, c f32 [N] = bias + w beta).  Once per weight."""
    wg = cast_f16((w * gamma[None, :]).contiguous())
    s = wg.float().sum(dim=1).contiguous()
    c = (w.double() @ beta.double()).float()
    if bias is not None:
        c = c + bias
    return wg, s, c.contiguous()


def row_stats_f16(x):
    """fp16 rows [..., C] -> f32 [rows, 2] = (mean, 1/sqrt(var + 1e-5)): the statistics of layernorm() without the normalised rows."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    _chk(x2)
    with _Prof("row_stats_h16_kernel", 0.0):
        return _mi().row_stats_f16(x2)


def gemm_nt_f16_res16_stats(a, w, bias, residual):
    """gemm_nt_f16 with an fp16 residual (out-proj / c_proj on the fp16 stream) that also returns the LayerNorm statistics of the rows
    it wrote: (out fp16 [M,N], stats f32 [M,2] = (mean, rstd)) -- partial sums from the GEMM's epilogue, finished by a 12-term sum per
    row; None instead of stats when the ping-pong kernel does not take the problem (callers then use row_stats_f16)."""
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    r2 = residual.reshape(-1, N)
    _chk(a2, w, bias, r2)
    M = a2.shape[0]
    if not (N % 256 == 0 and w.is_contiguous() and r2.dtype == torch.float16 and _lib.lib().ovis_gemm_nt_f16_res16_eligible(
            _lib._conv(r2), _lib._conv(r2), _ll(K), _ll(K), _ll(N), _ll(N), M, N, K, _lib._conv(bias))):
        return gemm_nt_f16(a, w, bias, residual), None
    with _Prof("gemm_f16_pp_kernel<1,0,true,false,false,true>", 2.0 * M * N * K):
        out, part = _mi().gemm_nt_f16_res16_stats(a2, w, bias, r2)
    return out.view(*a.shape[:-1], N), _mi().row_stats_finalize(part, N)


def gemm_nt_f16_ln_eligible(M, N, K, act=ACT_NONE):
    """whether gemm_nt_f16_ln takes this problem (otherwise: layernorm(out_f16=True) + gemm_nt_f16)."""
    big = 16 * 1024 * 1024          # alignment-only probes: the library checks pointers for 16-byte alignment, never reads them
    return bool(_lib.lib().ovis_gemm_nt_f16_ln_eligible(ctypes.c_void_p(big), _ll(K), _ll(K), _ll(N), int(M), int(N), int(K), ctypes.c_void_p(big),
                                                        ctypes.c_void_p(big), ctypes.c_void_p(big), int(act)))


def gemm_nt_f16_ln(x, wg, s, c, stats, act=ACT_NONE):
    """fp16( act( LayerNorm(x) w^T + b ) ) from the raw fp16 rows x [..., K], the folded weight (fold_layernorm) and the row statistics
    (row_stats_f16): rstd * (x wg^T - mean s) + c with f32 accumulation -- the normalised rows never exist in memory."""
    K = x.shape[-1]
    N = wg.shape[0]
    x2 = x.reshape(-1, K)
    _chk(x2, wg, s, c, stats)
    M = x2.shape[0]
    with _Prof("gemm_f16_pp_kernel<1,%d,false,false,false,false,256,5,true>" % act, 2.0 * M * N * K):
        out = _mi().gemm_nt_f16_ln(x2, wg, s, c, stats, int(act))
    return out.view(*x.shape[:-1], N)


def gemm_nt_batched(a, b, out, batch, M, N, K, lda, a_bs, ldb, b_bs, ldc, c_bs, bias=None, act=ACT_NONE, b16=None):
    """batch independent problems out_z = a_z b_z^T (pointer + z*stride, strides in elements); b16: fp16 copy of b
    (same strides) selects the autocast arithmetic.  Tensors give the base pointers only."""
    for t in (a, b, out, b16):
        if t is not None and not t.is_cuda:
            raise _lib.OvisError("gemm_nt_batched needs HIP tensors")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    if b16 is not None and K % 8 == 0:
        _lib.call("ovis_gemm_nt_f32a_f16w_batched", vp(a), _ll(lda), _ll(a_bs), vp(b16), _ll(ldb), _ll(b_bs), vp(out), _ll(ldc),
                  _ll(c_bs), batch, M, N, K, bias, act, _lib.stream_ptr())
    else:
        _lib.call("ovis_gemm_nt_f32_batched", vp(a), _ll(lda), _ll(a_bs), vp(b), _ll(ldb), _ll(b_bs), vp(out), _ll(ldc),
                  _ll(c_bs), batch, M, N, K, bias, act, _lib.stream_ptr())
    return out


def patch_row_len(patch):
    """row length of the ViT patch-embedding im2col matrix: 3*patch*patch padded to a multiple of 8 (16-byte fp16 rows;
    ViT-L/14: 588 -> 592, the pad columns are zero and meet zero weight columns)."""
    return (3 * patch * patch + 7) // 8 * 8


def _patch_matrix(rows, patch, out_f16, device):
    k, ld = 3 * patch * patch, patch_row_len(patch)
    dt = torch.float16 if out_f16 else torch.float32
    return torch.empty((rows, ld), dtype=dt, device=device) if ld == k else torch.zeros((rows, ld), dtype=dt, device=device)


def san_front_patches(frames, Hp, Wp, resolution, patch, mean, std, out_f16=False):
    _chk(frames)
    T, _, H, W = frames.shape
    G = resolution // patch
    A = _patch_matrix(T * G * G, patch, out_f16, frames.device)
    _lib.call("ovis_san_front_patches", frames, A, int(out_f16), T, H, W, Hp, Wp, resolution, patch, _ll(A.shape[1]), _f3(mean),
              _f3(std), _lib.stream_ptr())
    return A


def adaptive_maxpool2d(x, OH, OW):
    """x [..., H, W] -> [..., OH, OW]."""
    _chk(x)
    H, W = x.shape[-2:]
    y = torch.empty((*x.shape[:-2], OH, OW), dtype=torch.float32, device=x.device)
    _lib.call("ovis_adaptive_maxpool2d_f32", x, y, _ll(x.numel() // (H * W)), H, W, OH, OW, _lib.stream_ptr())
    return y


def san_attn_bias(pooled, Q, L):
    """pooled [B,n,Q,L] -> additive bias [B,n,Q+1+L,ld] (rows padded to a multiple of 4)."""
    _chk(pooled)
    B, n = pooled.shape[:2]
    S = Q + 1 + L
    ld = (S + 3) // 4 * 4
    out = torch.empty((B, n, S, ld), dtype=torch.float32, device=pooled.device)
    _lib.call("ovis_san_attn_bias_f32", pooled, out, _ll(B * n), Q, L, ld, _lib.stream_ptr())
    return out


def bilinear_resize_add(dst, src):
    """dst [N,H,W,C] += bilinear_resize(src [N,h,w,C]) in place."""
    _chk(dst, src)
    N, H, W, C = dst.shape
    _lib.call("ovis_bilinear_resize_add_nhwc_f32", dst, src, N, H, W, C, src.shape[1], src.shape[2], _lib.stream_ptr())
    return dst


def hungarian_link(embeds):
    """embeds f32 [T,Q,C] -> int32 indices [T,Q] (minvis.py:28-72 chain)."""
    _chk(embeds)
    return _mi().hungarian_link(embeds)


def batch_index_rows(src, idx, out, src_bs, src_rs, out_bs, out_rs, length):
    """out[b,m,:] = src[b, idx[b,m], :] with explicit element strides (see include/openvis_hip.h)."""
    B, M = idx.shape
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("ovis_batch_index_rows_f32", vp(src), _ll(src_bs), _ll(src_rs), idx, vp(out), _ll(out_bs), _ll(out_rs), B, M,
              _ll(length), _lib.stream_ptr())
    return out


def cast_f16_to_f32_rows(x, C=None):
    """fp16 rows -> dense f32 [rows, C]; x may be a strided row view (e.g. the class-token rows x[:, 0, :] of [B, L, C])."""
    if not (x.is_cuda and x.dtype == torch.float16 and x.dim() == 2 and x.stride(1) == 1):
        raise _lib.OvisError("cast_f16_to_f32_rows needs a 2-d fp16 HIP tensor with contiguous rows")
    C = x.shape[1] if C is None else C
    y = torch.empty((x.shape[0], C), dtype=torch.float32, device=x.device)
    _lib.call("ovis_cast_f16_to_f32_rows", ctypes.c_void_p(x.data_ptr()), _ll(x.stride(0)), y, _ll(x.shape[0]), C, _lib.stream_ptr())
    return y


def cast_f16(x):
    _chk(x)
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _lib.call("ovis_cast_f32_to_f16", x, y, _ll(x.numel()), _lib.stream_ptr())
    return y


def conv2d_nhwc(x, w, stride=1, pad=0, bias=None, residual=None, act=ACT_NONE, w16=None, cw=False):
    """x [N,H,W,Cin], w [Cout,KH,KW,Cin] -> [N,OH,OW,Cout]. w16: fp16 copy of w -> autocast arithmetic.
    cw=True: constant weights -> cached bf16x3 planes for the f32 path (see gemm_nt)."""
    _chk(x, w, bias, residual, w16)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device)
    if w16 is not None and (KH * KW * Cin) % 8 == 0:
        big = ((N * OH * OW + 127) // 128) * ((Cout + 127) // 128) >= 1024 and Cout > 64
        with _Prof(f"gemm_f16cvt_kernel<{'128,128' if big else '64,64'},ConvA>", 2.0 * N * OH * OW * Cout * KH * KW * Cin):
            _lib.call("ovis_conv2d_nhwc_f32a_f16w", x, w16, y, N, H, W, Cin, Cout, KH, KW, stride, pad, bias, residual, act,
                      _lib.stream_ptr())
        return y
    if cw and (KH * KW * Cin) % 8 == 0 and _MODE.v == 3 and ((N * OH * OW + 127) // 128) * ((Cout + 127) // 128) >= 256:     # fp16x2
        h2, ws = h2_of(w)
        label = ""
        if PROFILE is not None:
            label = _gemm_variant(N * OH * OW, Cout, "ConvA", KH * KW * Cin, h2=True)
            if KH == 1 and KW == 1 and stride == 1 and pad == 0:
                label = _h2_kernel_name(x, Cin, h2, Cin, w.numel(), y, Cout, N * OH * OW, Cout, Cin, bias, residual, Cout, act) or label
        with _Prof(label, 2.0 * N * OH * OW * Cout * KH * KW * Cin):
            _lib.call("ovis_conv2d_nhwc_f32_h2", x, w, h2, _ll(w.numel()), ctypes.c_float(ws), y, N, H, W, Cin, Cout, KH, KW, stride, pad, bias,
                      residual, act, _lib.stream_ptr())
        return y
    label = _gemm_variant(N * OH * OW, Cout, "ConvA", KH * KW * Cin) if PROFILE is not None else ""
    if (PROFILE is not None and cw and _MODE.v == 2 and KH == 1 and KW == 1 and stride == 1 and pad == 0 and Cin % 8 == 0
            and ((N * OH * OW + 127) // 128) * ((Cout + 127) // 128) >= 256):
        label = _w3_kernel_name(x, Cin, w3_of(w), Cin, w.numel(), y, Cout, N * OH * OW, Cout, Cin, bias, residual, Cout, act) or label
    with _Prof(label, 2.0 * N * OH * OW * Cout * KH * KW * Cin):
        if cw and (KH * KW * Cin) % 8 == 0 and _MODE.v >= 1:
            _lib.call("ovis_conv2d_nhwc_f32_w3", x, w, w3_of(w), _ll(w.numel()), y, N, H, W, Cin, Cout, KH, KW, stride, pad, bias,
                      residual, act, _lib.stream_ptr())
        else:
            _lib.call("ovis_conv2d_nhwc_f32", x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, bias, residual, act,
                      _lib.stream_ptr())
    return y


# ---- fp16 storage between the convolutions of a ResNet bottleneck (csrc/conv_h16.hip, gemm_f16cvt.hip; include/openvis_hip.h) -----------
def conv_h16(x16, w16, ksize, stride=1, bias=None, residual=None, act=ACT_NONE, out_f16=True):
    """x16 fp16 [T,H,W,Cin], w16 fp16 [Cout,k,k,Cin] (k = 1 or 3, pad k // 2) -> fp16 or f32 [T,OH,OW,Cout] = act(conv + bias (+ f32 residual))."""
    _chk(x16, w16, bias, residual)
    if x16.dtype != torch.float16 or w16.dtype != torch.float16:
        raise _lib.OvisError("conv_h16 needs fp16 activations and weights")
    T, H, W, Cin = x16.shape
    Cout = w16.shape[0]
    pad = ksize // 2
    OH, OW = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    y = torch.empty((T, OH, OW, Cout), dtype=torch.float16 if out_f16 else torch.float32, device=x16.device)
    with _Prof(f"conv_h16_kernel<{128 if Cout % 128 == 0 else 64},{ksize * ksize},{'true' if out_f16 else 'false'}>", 2.0 * T * OH * OW * Cout * ksize * ksize * Cin):
        _lib.call("ovis_conv_h16", x16, w16, y, int(out_f16), T, H, W, Cin, Cout, ksize, stride, bias, residual, act, _lib.stream_ptr())
    return y


def gemm_nt_x16(a, w16, bias=None, residual=None, act=ACT_NONE, out_f16=False):
    """a f32 or fp16 [..., K] x w16 fp16 [N, K] -> f32 (+ f32 residual) or fp16 [..., N]; fp16 MFMA operands, f32 accumulation (gemm_nt with
    w16, plus the fp16 storage of the operands / result)."""
    K = a.shape[-1]
    N = w16.shape[0]
    a2 = a.reshape(-1, K)
    _chk(a2, w16, bias, residual)
    M = a2.shape[0]
    out = torch.empty((M, N), dtype=torch.float16 if out_f16 else torch.float32, device=a.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    big = ((M + 127) // 128) * ((N + 127) // 128) >= 1024 and N > 64
    a_h = a2.dtype == torch.float16
    with _Prof(f"gemm_f16cvt_kernel<{'128,128' if big else '64,64'},{'DenseH' if a_h else 'DenseA'}{',o16' if out_f16 else ''}>", 2.0 * M * N * K):
        _lib.call("ovis_gemm_nt_x16", a2, int(a_h), _ll(K), w16, _ll(K), out, int(out_f16), _ll(N), M, N, K, bias, r2, _ll(N), act, _lib.stream_ptr())
    return out.view(*a.shape[:-1], N)


def gemm_nt_x16_2a(a1, a2, w16, bias=None, act=ACT_NONE):
    """[a1 | a2] w16^T + bias with a1 [..., K1], a2 [..., K2] fp16 and w16 fp16 [N, K1 + K2] -> f32 [..., N]: conv3 + projection shortcut of
    res2.0 as one GEMM (ovis_gemm_nt_x16_2a)."""
    K1, K2, N = a1.shape[-1], a2.shape[-1], w16.shape[0]
    x1, x2 = a1.reshape(-1, K1), a2.reshape(-1, K2)
    _chk(x1, x2, w16, bias)
    M = x1.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a1.device)
    big = ((M + 127) // 128) * ((N + 127) // 128) >= 1024 and N > 64
    with _Prof(f"gemm_f16cvt_kernel<{'128,128' if big else '64,64'},DualA>", 2.0 * M * N * (K1 + K2)):
        _lib.call("ovis_gemm_nt_x16_2a", x1, _ll(K1), K1, x2, _ll(K2), K2, w16, _ll(K1 + K2), out, _ll(N), M, N, bias, act, _lib.stream_ptr())
    return out.view(*a1.shape[:-1], N)


def conv1x1_pair_x16(a1, x2, stride, w16, bias=None, act=ACT_NONE):
    """a1 fp16 [T, OH, OW, K1] (conv2's output) and the f32 block input x2 [T, H, W, C2] at stride `stride` -> f32 [T, OH, OW, N]:
    conv3 + the strided 1x1 shortcut of res3.0 / res4.0 / res5.0 as one GEMM over [K1 | C2] (ovis_conv1x1_pair_x16)."""
    _chk(a1, x2, w16, bias)
    T, OH, OW, K1 = a1.shape
    _, H, W, C2 = x2.shape
    N = w16.shape[0]
    if (OH, OW) != ((H - 1) // stride + 1, (W - 1) // stride + 1) or w16.shape[1] != K1 + C2:
        raise _lib.OvisError("conv1x1_pair_x16: shapes of the two sources / the weight do not match")
    y = torch.empty((T, OH, OW, N), dtype=torch.float32, device=a1.device)
    M = T * OH * OW
    big = ((M + 127) // 128) * ((N + 127) // 128) >= 1024 and N > 64
    with _Prof(f"gemm_f16cvt_kernel<{'128,128' if big else '64,64'},DualA>", 2.0 * M * N * (K1 + C2)):
        _lib.call("ovis_conv1x1_pair_x16", a1, K1, x2, T, H, W, C2, stride, w16, y, N, bias, act, _lib.stream_ptr())
    return y


def resnet_stem_pool(x, w16, bias):
    """ResNet stem (7x8-tap conv, stride 2, folded FrozenBN, ReLU) + 3x3 / stride 2 max pool in one launch: x f32 [T, H, W, 4] -> fp16
    [T, PH, PW, 64] (ovis_resnet_stem_pool_f16)."""
    _chk(x, w16, bias)
    T, H, W, C = x.shape
    if C != 4 or tuple(w16.shape) != (64, 7, 8, 4) or w16.dtype != torch.float16:
        raise _lib.OvisError("resnet_stem_pool: x must be NHWC4 and w16 fp16 [64, 7, 8, 4]")
    OH, OW = (H - 1) // 2 + 1, W // 2
    y = torch.empty((T, (OH - 1) // 2 + 1, (OW - 1) // 2 + 1, 64), dtype=torch.float16, device=x.device)
    with _Prof("stem_pool_kernel", 2.0 * T * OH * OW * 64 * 224):
        _lib.call("ovis_resnet_stem_pool_f16", x, w16, bias, y, T, H, W, _lib.stream_ptr())
    return y


def conv2d_nhwc_o16(x, w16, stride, pad, bias=None, act=ACT_NONE):
    """conv2d_nhwc with fp16 weights (autocast arithmetic) writing an fp16 map (the ResNet stem of the fp16-storage backbone)."""
    _chk(x, w16, bias)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w16.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = torch.empty((N, OH, OW, Cout), dtype=torch.float16, device=x.device)
    big = ((N * OH * OW + 127) // 128) * ((Cout + 127) // 128) >= 1024 and Cout > 64
    with _Prof(f"gemm_f16cvt_kernel<{'128,128' if big else '64,64'},ConvA,o16>", 2.0 * N * OH * OW * Cout * KH * KW * Cin):
        _lib.call("ovis_conv2d_nhwc_f32a_f16w_o16", x, w16, y, N, H, W, Cin, Cout, KH, KW, stride, pad, bias, act, _lib.stream_ptr())
    return y


# ---------------------------------------------------------------------------------------------
def _f3(vals):
    return (ctypes.c_float * 3)(*[float(v) for v in vals])


def preprocess_u8(frames, Hp, Wp, mean, std):
    """frames uint8 [T,3,H,W] -> f32 [T,Hp,Wp,4] normalised, zero padded (openvis.py:57-62)."""
    _chk(frames)
    T, _, H, W = frames.shape
    out = torch.empty((T, Hp, Wp, 4), dtype=torch.float32, device=frames.device)
    _lib.call("ovis_preprocess_u8_nhwc4", frames, out, T, H, W, Hp, Wp, _f3(mean), _f3(std), _lib.stream_ptr())
    return out


def maxpool3x3s2(x):
    _chk(x)
    N, H, W, C = x.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    _lib.call("ovis_maxpool3x3s2_nhwc_f16" if x.dtype == torch.float16 else "ovis_maxpool3x3s2_nhwc_f32", x, y, N, H, W, C, _lib.stream_ptr())
    return y


def layernorm(x, gamma, beta, residual=None, eps=1e-5, out_f16=False):
    _chk(x, gamma, beta, residual)
    C = x.shape[-1]
    y = torch.empty(x.shape, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    if x.dtype == torch.float16:             # fp16 residual stream: statistics in f32 (CLIP's LayerNorm subclass, model.py:157-163)
        if residual is not None:
            raise _lib.OvisError("layernorm: no residual input with an fp16 x")
        _lib.call("ovis_layernorm_f16_to_f16" if out_f16 else "ovis_layernorm_f16_to_f32", x, gamma, beta, y, _ll(x.numel() // C), C,
                  float(eps), _lib.stream_ptr())
        return y
    _lib.call("ovis_layernorm_f32_to_f16" if out_f16 else "ovis_layernorm_f32", x, residual, gamma, beta, y,
              _ll(x.numel() // C), C, float(eps), _lib.stream_ptr())
    return y


def groupnorm_nhwc(x, gamma, beta, groups=32, eps=1e-5, relu=False, up_add=None, pad=False):
    """GroupNorm of an NHWC map (+ bilinear-upsampled addend, + ReLU).  pad=True: the result is written into a zero-padded map
    [N, H+2, W+2, C] (ring of zeros stored by the same kernel) -- the input layout of conv3x3_padded."""
    _chk(x, gamma, beta, up_add)
    N, H, W, C = x.shape
    y = torch.empty((N, H + 2, W + 2, C), dtype=torch.float32, device=x.device) if pad else torch.empty_like(x)
    ws = torch.empty((N * groups * 2,), dtype=torch.float64, device=x.device)
    UH, UW = (up_add.shape[1], up_add.shape[2]) if up_add is not None else (0, 0)
    _lib.call("ovis_groupnorm_nhwc_f32_padded" if pad else "ovis_groupnorm_nhwc_f32", x, y, gamma, beta, ws, N, H, W, C, groups, float(eps),
              int(relu), up_add, UH, UW, _lib.stream_ptr())
    return y


def conv3x3_padded_eligible(N, H, W, Cin, Cout, act=ACT_NONE):
    """whether conv3x3_padded takes this problem on the ping-pong kernel (bf16x2 policy, Cin % 32 == 0, enough tiles); alignment-only
    probe pointers (the library checks 16-byte alignment, it does not read them)"""
    big = ctypes.c_void_p(16 * 1024 * 1024)
    if _MODE.v == 3:
        return bool(_lib.lib().ovis_conv3x3_padded_f32_h2_eligible(big, big, _ll(Cout * 9 * Cin), big, int(N), int(H), int(W), int(Cin), int(Cout),
                                                                   None, int(act)))
    return _MODE.v == 2 and bool(_lib.lib().ovis_conv3x3_padded_f32_w3_eligible(big, big, _ll(Cout * 9 * Cin), big, int(N), int(H), int(W), int(Cin),
                                                                                  int(Cout), None, int(act)))


def conv3x3_padded(xpad, w, bias=None, act=ACT_NONE):
    """3x3 / stride 1 / pad 1 convolution of an NHWC map given ZERO-PADDED: xpad [N, H+2, W+2, Cin] (groupnorm_nhwc(pad=True)), constant
    weight w [Cout, 3, 3, Cin] -> [N, H, W, Cout].  The ping-pong kernel walks the padded map as a dense GEMM (no im2col gather); where it
    does not take the problem, the ordinary convolution runs on the interior."""
    _chk(xpad, w, bias)
    N, Hp2, Wp2, Cin = xpad.shape
    H, W = Hp2 - 2, Wp2 - 2
    Cout = w.shape[0]
    if tuple(w.shape[1:]) != (3, 3, Cin):
        raise _lib.OvisError("conv3x3_padded: w must be [Cout, 3, 3, Cin]")
    if conv3x3_padded_eligible(N, H, W, Cin, Cout, act):
        y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=xpad.device)
        if _MODE.v == 3:
            h2, ws = h2_of(w)
            with _Prof("gemm_f16_pp_kernel<0,%d,false,false,true,false,FH>" % act, 2.0 * N * H * W * Cout * 9 * Cin):
                _lib.call("ovis_conv3x3_padded_f32_h2", xpad, h2, _ll(w.numel()), ctypes.c_float(ws), y, N, H, W, Cin, Cout, bias, int(act),
                          _lib.stream_ptr())
            return y
        with _Prof("gemm_f16_pp_kernel<0,%d,false,false,true,false>" % act, 2.0 * N * H * W * Cout * 9 * Cin):
            _lib.call("ovis_conv3x3_padded_f32_w3", xpad, w3_of(w), _ll(w.numel()), y, N, H, W, Cin, Cout, bias, int(act), _lib.stream_ptr())
        return y
    return conv2d_nhwc(xpad[:, 1:-1, 1:-1, :].contiguous(), w, 1, 1, bias=bias, act=act, cw=True)


def ln_mlp3(x, gamma, beta, wts, biases, want_dec=True, eps=1e-5):
    """decoder_norm + 3-layer MLP (ReLU between) in one launch; wts = the three Linear weights TRANSPOSED [in, out].  Returns (dec, out)."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    _chk(x2, gamma, beta, *wts, *biases)
    out = torch.empty_like(x2)
    dec = torch.empty_like(x2) if want_dec else None
    _lib.call("ovis_ln_mlp3_f32", x2, gamma, beta, wts[0], biases[0], wts[1], biases[1], wts[2], biases[2], dec, out, x2.shape[0], C,
              ctypes.c_float(eps), _lib.stream_ptr())
    return (dec.view(x.shape) if want_dec else None), out.view(x.shape)


def add_bcast(a, b):
    """a + b with b broadcast over the leading dims of a (b.numel() divides a.numel())."""
    _chk(a, b)
    out = torch.empty_like(a)
    _lib.call("ovis_add_bcast_f32", a, b, out, _ll(a.numel()), _ll(b.numel()), _lib.stream_ptr())
    return out


def pe_sine(T, H, W, npf, three_d, add_c, device):
    out = torch.empty((T, H, W, 2 * npf), dtype=torch.float32, device=device)
    _chk(add_c)
    _lib.call("ovis_pe_sine_f32", out, T, H, W, npf, int(three_d), add_c, _lib.stream_ptr())
    return out


def attention(q, k, v, B, H, Nq, Nk, D, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, mask=None, row_open=None, nsplit=1,
              out_f16=False, mask_per_batch=False, bias=None, out=None, o_bs=None, o_ld=None, bias_strides=None):
    """q/k/v: tensors (possibly column-sliced views of a fused projection) whose element (b,row,h,d) sits at
    data_ptr + (b*bs + row*ld + h*D + d)*4.  Returns out [B,Nq,H*D]."""
    for t in (q, k, v):
        if not t.is_cuda:
            raise _lib.OvisError("attention needs HIP tensors")
    if out is None:
        out = torch.empty((B, Nq, H * D), dtype=torch.float16 if out_f16 else torch.float32, device=q.device)
        o_bs, o_ld = Nq * H * D, H * D
    ws = None
    if nsplit > 1:
        nbytes = _lib.lib().ovis_attention_workspace_bytes(B, H, Nq, D, nsplit)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=q.device)
    mask_ld = mask.shape[-1] if mask is not None else 0
    mask_bs = Nq * mask_ld if (mask is not None and mask_per_batch) else 0      # mask [B*Nq, ld] when per batch
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    if bias is not None:                      # additive f32 bias [B,H,Nq,ld]
        _chk(bias)
        b_ld = bias.shape[-1]
        bs_, hs_ = bias_strides if bias_strides is not None else (H * Nq * b_ld, Nq * b_ld)   # (batch, head) strides
        b_args = (bias, _ll(bs_), _ll(hs_), b_ld)
    else:
        b_args = (None, _ll(0), _ll(0), 0)
    _lib.call("ovis_attention_f32", vp(q), _ll(q_bs), q_ld, vp(k), _ll(k_bs), k_ld, vp(v), _ll(v_bs), v_ld, out,
              _ll(o_bs), o_ld, int(out_f16), mask, _ll(mask_ld), _ll(mask_bs), row_open, *b_args, B, H, Nq, Nk, D,
              float(D) ** -0.5, nsplit, ws, _lib.stream_ptr())
    return out


def attention_partial(q, k, v, B, H, Nq, Nk, D, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, mask=None, row_open=None, nsplit=1, mask_per_batch=False):
    """This GPU's share of an attention whose KEYS are split over GPUs (include/openvis_hip.h: ovis_attention_partial_f32): the packed
    un-normalised partial, f32 [ovis_attention_partial_floats(B,H,Nq,D)].  All-gather the blocks, then attention_merge."""
    for t in (q, k, v):
        if not t.is_cuda:
            raise _lib.OvisError("attention needs HIP tensors")
    L = _lib.lib()
    part = torch.empty((L.ovis_attention_partial_floats(B, H, Nq, D),), dtype=torch.float32, device=q.device)
    ws = torch.empty((L.ovis_attention_partial_workspace_bytes(B, H, Nq, D, nsplit) // 4,), dtype=torch.float32, device=q.device)
    mask_ld = mask.shape[-1] if mask is not None else 0
    mask_bs = Nq * mask_ld if (mask is not None and mask_per_batch) else 0
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("ovis_attention_partial_f32", vp(q), _ll(q_bs), q_ld, vp(k), _ll(k_bs), k_ld, vp(v), _ll(v_bs), v_ld, mask, _ll(mask_ld),
              _ll(mask_bs), row_open, B, H, Nq, Nk, D, float(D) ** -0.5, nsplit, ws, part, _lib.stream_ptr())
    return part


def attention_merge(parts, B, H, Nq, D):
    """parts f32 [R, n] (n >= ovis_attention_partial_floats): the gathered blocks of R GPUs -> out [B, Nq, H*D]."""
    _chk(parts)
    R, n = parts.shape
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=parts.device)
    _lib.call("ovis_attention_merge_f32", parts, R, _ll(n), out, _ll(Nq * H * D), H * D, B, H, Nq, D, _lib.stream_ptr())
    return out


def attention_f16(q, k, v, B, H, Nq, Nk, D, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld):
    """fp16 q/k/v views (element (b,row,h,d) at data_ptr + (b*bs + row*ld + h*D + d)*2) -> fp16 out [B,Nq,H*D]."""
    for t in (q, k, v):
        if not (t.is_cuda and t.dtype == torch.float16):
            raise _lib.OvisError("attention_f16 needs fp16 HIP tensors")
    return _mi().attention_f16(q, k, v, B, H, Nq, Nk, D, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld)


# LDS-staged tiled K1 (csrc/openvis_ops.hip: msda_encoder_tiled_kernel): radius in pixels of the value window staged per
# 8x8 query tile, or -1 = off.  OFF by default: measured on MI355X (tools/bench_msda_fused.py, 5 frames at 720p) the
# direct-gather kernel takes 370 us, the tiled one 480 us (radius 2-3) / 720 us (radius 4): the windows' halo makes the
# staged volume ~5x the tile's own pixels, more than what L1/L2 already save the direct gather.
MSDA_TILE_RADIUS = -1


def mask_bbox_set_cells(on):
    """lab / tests: False = the per-pixel mask_bbox_kernel for every size; True (default) = mask_bbox4_kernel when the masks are at stride 4."""
    _lib.call("ovis_mask_bbox_set_cells", int(bool(on)))


def msda_set_share(on):
    """lab / tests: False = msda_encoder_fused_kernel (every lane computes every sampling point); True (default) = the lane-sharing
    msda_encoder_fused8_kernel (bit-identical output)."""
    _lib.call("ovis_msda_set_share", int(bool(on)))


def msda_encoder_fused(value, oa, shapes, lsi, M=8, L=3, P=4, shapes_host=None):
    """value [B,S,C], oa [B,S,M*L*P*3] -> [B,S,C].  shapes_host: the level shapes as python ints [(H,W)] * L (coarse to
    fine) -> the LDS-staged tiled kernel handles the finest level's queries."""
    _chk(value, oa, shapes, lsi)
    B, S, C = value.shape
    # algorithmic bytes (DESIGN.md section 3 / SURVEY.md 8d): f32 value + offsets/logits + output rows, once each
    nbytes = 4.0 * B * S * (2 * C + oa.shape[-1])
    if shapes_host is not None and L == 3 and P == 4 and C // M == 32 and MSDA_TILE_RADIUS >= 0:
        sh = (ctypes.c_int * 6)(*[int(v) for hw in shapes_host for v in hw])
        out = torch.empty_like(value)
        with _Prof(f"msda_encoder_tiled_kernel<{P}>+fused<{L},{P}>", nbytes, unit="byte"):
            _lib.call("ovis_msda_encoder_fused_tiled_f32", value, oa, oa.shape[-1], shapes, lsi, sh, out, B, S, M, C // M, L, P,
                      MSDA_TILE_RADIUS, _lib.stream_ptr())
        return out
    kname = "msda_encoder_fused8_kernel" if C // M == 32 else "msda_encoder_fused_kernel"      # head_dim 32: the lane-sharing kernel
    with _Prof(f"{kname}<{L},{P}>", nbytes, unit="byte"):
        out = _mi().msda_encoder_fused(value, oa, shapes, lsi, M, L, P)
    return out


def attn_mask_from_logits(logits):
    """logits [Q,Nk] -> (mask uint8 [Q,Nk4], row_open int32 [Q])."""
    _chk(logits)
    Q, Nk = logits.shape
    ld = (Nk + 3) // 4 * 4
    mask = torch.empty((Q, ld), dtype=torch.uint8, device=logits.device)
    row_open = torch.empty((Q,), dtype=torch.int32, device=logits.device)
    _lib.call("ovis_attn_mask_from_logits", logits, _ll(Nk), mask, _ll(ld), row_open, Q, Nk, _lib.stream_ptr())
    return mask, row_open


def center_pool(x, s):
    _chk(x)
    N, H, W, C = x.shape
    y = torch.empty((N, H // s, W // s, C), dtype=torch.float32, device=x.device)
    _lib.call("ovis_center_pool_nhwc_f32", x, y, N, H, W, C, s, _lib.stream_ptr())
    return y


def mask_bbox(masks, Hp, Wp):
    """masks [Q,T,h,w] logits -> int32 [T,Q,4] inclusive boxes at (Hp,Wp) resolution (x1 < 0: empty)."""
    _chk(masks)
    boxes = _mi().mask_bbox(masks, int(Hp), int(Wp))
    return boxes


def crop_list_static(boxes, Hp, Wp):
    """boxes int32 [T,Q,4] (device) -> (crops int32 [T*Q,6], slot int32 [T,Q], counts int32 [1] = number of non-empty masks), all on the
    device and of data-independent shape (include/openvis_hip.h: ovis_crop_list_static)."""
    _chk(boxes)
    T, Q = boxes.shape[:2]
    crops = torch.empty((T * Q, 6), dtype=torch.int32, device=boxes.device)
    slot = torch.empty((T, Q), dtype=torch.int32, device=boxes.device)
    counts = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    _lib.call("ovis_crop_list_static", boxes, crops, slot, counts, T, Q, int(Hp), int(Wp), _lib.stream_ptr())
    return crops, slot, counts


def clip_crop_patches(frames, masks, crops, Hp, Wp, resolution, patch, mean, std, out_f16=False):
    _chk(frames, masks, crops)
    return _mi().clip_crop_patches(frames, masks, crops, int(Hp), int(Wp), int(resolution), int(patch), [float(v) for v in mean],
                                   [float(v) for v in std], bool(out_f16))


def clip_crop_patches_masked(frames, masks, crops, Hp, Wp, resolution, patch, mean, std, out_f16=False):
    """AdaptedClipAdapter crops: (patch matrix A, patch_open uint8 [M, G*G]) — see ovis_clip_crop_patches_masked."""
    _chk(frames, masks, crops)
    T, _, H, W = frames.shape
    Q, _, h, w = masks.shape
    M = crops.shape[0]
    G = resolution // patch
    A = _patch_matrix(M * G * G, patch, out_f16, frames.device)
    patch_open = torch.empty((M, G * G), dtype=torch.uint8, device=frames.device)
    wsb = int(_lib.lib().ovis_clip_crop_workspace_bytes(M, resolution))
    ws = torch.empty(((wsb + 3) // 4,), dtype=torch.float32, device=frames.device)       # leader / follower passes (csrc/openvis_ops.hip)
    _lib.call("ovis_clip_crop_patches_ws", frames, masks, crops, A, patch_open, int(out_f16), M, Q, T, H, W, h, w, Hp, Wp,
              resolution, patch, _ll(A.shape[1]), _f3(mean), _f3(std), ws, _ll(wsb), _lib.stream_ptr())
    return A, patch_open


def mask_prompt_select(x, patch_open, mask_embedding, first_token):
    """In place: closed patch tokens of x f32 [M, tokens, C] take mask_embedding [1 or L, C] (model.py:334-338, 349-352)."""
    _chk(x, patch_open, mask_embedding)
    M, L = patch_open.shape
    if x.dim() == 2:
        x = x.view(M, L, -1)
    assert x.is_contiguous() and x.dtype == torch.float32 and mask_embedding.is_contiguous()
    _lib.call("ovis_mask_prompt_select_f32", x, patch_open, mask_embedding, M, L, x.shape[2], x.shape[1], first_token,
              mask_embedding.shape[0], _lib.stream_ptr())
    return x


def vit_embed_ln(patch, cls, pos, gamma, beta, M, L1, eps=1e-5):
    _chk(patch, cls, pos, gamma, beta)
    C = cls.numel()
    if patch.dtype == torch.float16:         # fp16 patch embeddings in, fp16 tokens out (the tower's fp16 residual stream)
        out = torch.empty((M, L1, C), dtype=torch.float16, device=patch.device)
        _lib.call("ovis_vit_embed_ln_f16", patch, cls, pos, gamma, beta, out, M, L1, C, float(eps), _lib.stream_ptr())
        return out
    out = torch.empty((M, L1, C), dtype=torch.float32, device=patch.device)
    _lib.call("ovis_vit_embed_ln_f32", patch, cls, pos, gamma, beta, out, M, L1, C, float(eps), _lib.stream_ptr())
    return out


def l2norm_rows(x, scale=1.0):
    _chk(x)
    y = torch.empty_like(x)
    _lib.call("ovis_l2norm_rows_f32", x, y, _ll(x.shape[0]), x.shape[1], float(scale), _lib.stream_ptr())
    return y


def openvis_aggregate(crop_logits, slot, fill=0.0):
    """fill: value of the probability rows of queries without a valid crop (the kernel leaves them untouched): 0 (the compacted host path
    never selects them by row id) or -1 (device crop list: every row takes part in the top-k and must lose against any probability)."""
    _chk(crop_logits, slot)
    T, Q = slot.shape
    K = crop_logits.shape[1]
    probs = torch.full((Q, K), float(fill), dtype=torch.float32, device=slot.device)
    qvalid = torch.empty((Q,), dtype=torch.int32, device=slot.device)
    _lib.call("ovis_openvis_aggregate_f32", crop_logits, slot, probs, qvalid, T, Q, K, _lib.stream_ptr())
    return probs, qvalid


def topk_entropy(probs, row_ids, topk):
    _chk(probs, row_ids)
    return _mi().topk_entropy(probs, row_ids, int(topk))


def final_masks_set_cells(on):
    """lab / tests: False = the per-pixel kernels for every shape (the 4 x 4-cell kernel is bit-identical to them)"""
    _lib.call("ovis_final_masks_set_cells", int(bool(on)))


def final_masks(masks, sel_q, Hp, Wp, H, W, OH, OW, column_major=False):
    """-> uint8 [n,T,OH,OW] (or [n,T,OW,OH] column-major, the scan order of COCO RLE)."""
    _chk(masks, sel_q)
    Q, T, h, w = masks.shape
    n = sel_q.numel()
    out = torch.empty((n, T, OW, OH) if column_major else (n, T, OH, OW), dtype=torch.uint8, device=masks.device)
    _lib.call("ovis_final_masks_u8", masks, sel_q, out, n, Q, T, h, w, Hp, Wp, H, W, OH, OW, int(column_major), _lib.stream_ptr())
    return out


def rle_encode(masks_cm, cap=None):
    """masks_cm uint8 [n, len] (each mask flattened column-major) -> (counts int32 [n, cap], n_runs int32 [n]) on the device:
    uncompressed COCO RLE counts (see include/openvis_hip.h)."""
    _chk(masks_cm)
    n, length = masks_cm.shape
    cap = int(cap or length + 1)
    counts = torch.empty((n, cap), dtype=torch.int32, device=masks_cm.device)
    n_runs = torch.empty((n,), dtype=torch.int32, device=masks_cm.device)
    _lib.call("ovis_rle_encode_u8", masks_cm, n, _ll(length), counts, n_runs, cap, _lib.stream_ptr())
    return counts, n_runs


# ---- Swin backbone data movement (csrc/swin_ops.hip) -----------------------------------------------------------------
def swin_window_partition(x, ws, shift):
    """x [B,H,W,C] -> zero-padded, cyclically shifted windows [B*nW, ws*ws, C] (swin.py:241-262)."""
    _chk(x)
    B, H, W, C = x.shape
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    win = torch.empty((B * (Hp // ws) * (Wp // ws), ws * ws, C), dtype=x.dtype, device=x.device)
    # pure 16-byte data movement: an fp16 map is moved as C/2 "floats" per token
    cf = C if x.dtype == torch.float32 else C // 2
    if x.dtype == torch.float16 and C % 8:
        raise _lib.OvisError("swin_window_partition: fp16 maps need C % 8 == 0")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("ovis_swin_window_partition_f32", vp(x), vp(win), B, H, W, cf, ws, shift, _lib.stream_ptr())
    return win


def swin_window_merge_add(win, shortcut, ws, shift):
    """shortcut [B,H,W,C] + window_reverse / un-shift / crop of win [B*nW, ws*ws, C] (swin.py:267-281)."""
    _chk(win, shortcut)
    B, H, W, C = shortcut.shape
    out = torch.empty_like(shortcut)
    _lib.call("ovis_swin_window_merge_add_f32", win, shortcut, out, B, H, W, C, ws, shift, _lib.stream_ptr())
    return out


def swin_shift_mask(H, W, ws, shift, ld, device):
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    mask = torch.empty(((Hp // ws) * (Wp // ws), ws * ws, ld), dtype=torch.uint8, device=device)
    _lib.call("ovis_swin_shift_mask_u8", mask, H, W, ws, shift, ld, _lib.stream_ptr())
    return mask


def swin_patch_merge_gather(x):
    _chk(x)
    B, H, W, C = x.shape
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), dtype=torch.float32, device=x.device)
    _lib.call("ovis_swin_patch_merge_gather_f32", x, out, B, H, W, C, _lib.stream_ptr())
    return out


def swin_relpos_bias(table, heads, ws, ld):
    _chk(table)
    bias = torch.empty((heads, ws * ws, ld), dtype=torch.float32, device=table.device)
    _lib.call("ovis_swin_relpos_bias_f32", table, bias, heads, ws, ld, _lib.stream_ptr())
    return bias


def swin_window_attention_f16(qkv, bias, mask, heads):
    """qkv fp16 [nwin,N,3C] -> fp16 [nwin,N,C]; bias f32 [heads,N,ld]; mask u8 [nW,N,ld] (window = b mod nW) or None."""
    _chk(qkv, bias, mask)
    nwin, N, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty((nwin, N, C), dtype=torch.float16, device=qkv.device)
    _lib.call("ovis_swin_window_attention_f16", qkv, out, bias, mask, _ll(nwin), N, C, heads, mask.shape[0] if mask is not None else 0,
              bias.shape[-1], float(32) ** -0.5, _lib.stream_ptr())
    return out


def mean_over_dim0(x):
    """x [n, ...] -> mean over the first dim (prompt ensemble)."""
    _chk(x)
    y = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
    _lib.call("ovis_mean_dim0_f32", x, y, x.shape[0], _ll(y.numel()), _lib.stream_ptr())
    return y


# ---- f32-grade GEMM on pre-split bf16 planes (csrc/gemm_f16_pp.hip, X3 mode) ---------------------------------------------
def x3pp_eligible(M, N, K, has_bias=True):
    return bool(_lib.lib().ovis_gemm_x3pp_eligible(int(M), int(N), int(K), int(bool(has_bias))))


def split_planes(x):
    """f32 [..., K] -> bf16 [3, ..., K] with x == p0 + p1 + p2 exactly (the operand format of gemm_nt_planes)."""
    _chk(x)
    p = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    _lib.call("ovis_split_f32_to_bf16x3_v8", x, p, _ll(x.numel()), _lib.stream_ptr())
    return p


def gemm_nt_planes(a3, w3, bias=None, residual=None, act=ACT_NONE, out_planes=False):
    """a3 bf16 [3,M,K], w3 bf16 [3,N,K] (exact 3-way splits) -> f32 [M,N] or, out_planes, bf16 [3,M,N]: act(a w^T + bias + residual)
    with f32-grade accuracy on the ping-pong bf16 MFMA kernel (six plane-pair products as one K axis of 6 K)."""
    _chk(a3, w3, bias, residual)
    _, M, K = a3.shape
    N = w3.shape[1]
    out = torch.empty((3, M, N), dtype=torch.bfloat16, device=a3.device) if out_planes else \
        torch.empty((M, N), dtype=torch.float32, device=a3.device)
    with _Prof(f"gemm_f16_pp_kernel<{2 if out_planes else 0},{act},{'true' if residual is not None else 'false'},true,false,false>", 2.0 * M * N * K):
        _lib.call("ovis_gemm_nt_bf16x3_planes", a3, _ll(K), _ll(M * K), w3, _ll(K), _ll(N * K), out, _ll(N), _ll(M * N), M, N, K, bias,
                  residual, _ll(N), act, int(out_planes), _lib.stream_ptr())
    return out
