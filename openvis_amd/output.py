"""`video_output` of the eval forward (reference: video_maskformer.py:283-298, a plain dict) with a deferred device -> host
hand-off.

The forward ends with the D2H copy of the ten output masks (46 MB per 720p clip) and of the top-10 scalars.  Waiting for
that copy inside `forward` leaves the GPU idle until the host has come back, set up the next clip and uploaded its frames
(measured: 5-6 ms of a 46 ms step, tools/trace_gaps.py).  `VideoOutput` carries the copies' completion event instead: every
read of a device-produced field (`out["pred_masks"]`, `.items()`, ...) waits for it first, so the mapping behaves exactly like
the reference's dict, and a caller that enqueues the next clip before it reads the previous result gets the copy overlapped
with that clip's first stages -- the pattern of an evaluation loop (`for clip in loader: process(previous); previous =
model(clip)`), of `bench.py`, and of `runtime.ClipPipeline`.  The copy runs on a per-device side stream behind the last
kernel of the clip; the compute stream never waits for it.
"""
import threading
from collections.abc import MutableMapping

import torch

_copy_streams = {}
_lock = threading.Lock()


def copy_stream(device):
    """The per-device side stream of the output hand-off (HIP streams are cheap, but one is enough: copies serialise on PCIe)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    with _lock:
        s = _copy_streams.get(idx)
        if s is None:
            s = _copy_streams[idx] = torch.cuda.Stream(device=idx)
    return s


class VideoOutput(MutableMapping):
    def __init__(self, ready, event=None, finish=None):
        """ready: fields available now; finish() -> dict of the fields that need `event` (called once, after the wait)."""
        self._d = dict(ready)
        self._event, self._finish = event, finish
        self._pending = finish is not None

    def wait(self):
        """Block until the device -> host copies of this output have completed (no-op afterwards)."""
        if self._pending:
            if self._event is not None:
                self._event.synchronize()
            self._d.update(self._finish())
            self._pending, self._event, self._finish = False, None, None
        return self

    @property
    def pending(self):
        return self._pending

    def __getitem__(self, k):
        if self._pending and k not in self._d:
            self.wait()
        return self._d[k]

    def __setitem__(self, k, v):
        self._d[k] = v

    def __delitem__(self, k):
        self.wait()
        del self._d[k]

    def __iter__(self):
        self.wait()
        return iter(self._d)

    def __len__(self):
        self.wait()
        return len(self._d)

    def __contains__(self, k):
        if k in self._d:
            return True
        self.wait()
        return k in self._d

    def __repr__(self):
        return f"VideoOutput({'pending' if self._pending else repr(self._d)})"
