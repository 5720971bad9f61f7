"""ctypes binding of libopenvis_hip.so (the C-ABI drop-in boundary).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C openvis_amd/csrc``.  Loading fails loudly: there is no fallback path.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libopenvis_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "openvis_hip.h")

_lib = None


class OvisError(RuntimeError):
    pass


def declared_symbols(header_path=HEADER_PATH):
    """Names of every function declared in include/openvis_hip.h."""
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ovis_[a-z0-9_]+)\s*\(", src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OvisError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C openvis_amd/csrc`). openvis_amd has no CPU/eager fallback.")
        # torch bundles its own HIP runtime (soname libamdhip64.so.7); import it FIRST so this
        # library binds to that already-loaded runtime instead of pulling /opt/rocm's copy into
        # the process (two runtimes => "no ROCm-capable device" and foreign streams).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ovis_last_error.restype = ctypes.c_char_p
        _lib.ovis_abi_version.restype = ctypes.c_int
        for fn in ("ovis_attention_workspace_bytes", "ovis_hungarian_link_workspace_bytes", "ovis_clip_crop_workspace_bytes",
                   "ovis_attention_partial_floats", "ovis_attention_partial_workspace_bytes"):
            if hasattr(_lib, fn):
                getattr(_lib, fn).restype = ctypes.c_longlong
    return _lib


def _conv(a):
    """torch.Tensor -> device pointer; ints/floats pass through."""
    import torch
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, torch.Tensor):
        if not a.is_contiguous():
            raise OvisError("non-contiguous tensor passed to the C ABI")
        return ctypes.c_void_p(a.data_ptr())
    if isinstance(a, float):
        return ctypes.c_float(a)
    return a


# ---- bounded launch run-ahead --------------------------------------------------------------------------------------------------------
# LAUNCH_WINDOW > 0: the calling thread never has more than ~3 x LAUNCH_WINDOW library launches queued ahead of the GPU: every LAUNCH_WINDOW
# launches it records an event on its stream and waits for the event recorded two windows earlier (already complete in the steady state of
# a GPU-bound forward, so the wait costs nothing -- it only stops the host from running whole clips ahead).  Measured on the headline
# (profiles/r04/launch_window.txt): once the forward has no read-back in its middle (MODEL.CLIP_ADAPTER.CROP_LIST device) the host queues
# clips ahead without bound and the step gets 1.7 % SLOWER (131.9 frames/s against 133.8 with the read-back; kernel time unchanged, the
# loss is between the kernels); a window of 32-64 gives 134.3, a window of 128-256 133.2.  0 = unlimited (OVIS_LAUNCH_WINDOW overrides).
LAUNCH_WINDOW = int(os.environ.get("OVIS_LAUNCH_WINDOW", "64"))
_tls = __import__("threading").local()


def _throttle():
    n = getattr(_tls, "n", 0) + 1
    if n < LAUNCH_WINDOW:
        _tls.n = n
        return
    _tls.n = 0
    import torch
    ev = torch.cuda.Event()
    ev.record()
    q = getattr(_tls, "q", None)
    if q is None:
        q = _tls.q = []
    q.append(ev)
    if len(q) > 2:
        q.pop(0).synchronize()


def call(name, *args):
    """Call an int-returning ovis_* entry point; raise OvisError on a non-zero code."""
    fn = getattr(lib(), name)
    rc = fn(*[_conv(a) for a in args])
    if rc != 0:
        raise OvisError(f"{name} failed (code {rc}): {lib().ovis_last_error().decode()}")
    if LAUNCH_WINDOW > 0:
        _throttle()


_raw_stream = None


def stream_ptr():
    """Current torch HIP stream (of the calling thread's current device) as a void* for the ABI's `stream` argument.
    torch.cuda.current_stream() builds a Stream object through several Python layers (~8 us per call, 2.7 ms of host time per 720p clip
    at ~330 C-ABI launches); the two C accessors behind it give the same handle in well under a microsecond."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get is not None and dev is not None:
            _raw_stream = lambda: get(dev())
        else:                                                       # other torch builds: the public, slower path
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return ctypes.c_void_p(_raw_stream())
