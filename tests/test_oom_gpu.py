"""GPU: the out-of-memory ladder around the eval forward (openvis_amd/modeling/video_maskformer.py `retry_if_oom`), the counterpart of
detectron2's `retry_if_cuda_oom` that the reference wraps its big stages in (openvis.py:108, video_maskformer.py:205 / 213,
brivis.py:201 / 211): run -> empty the caching allocator and run again -> (per-frame models) run as windows -> re-raise."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _failing(fn, n_failures, calls):
    def wrapped(*a, **k):
        calls.append(a[0].shape[0] if a and torch.is_tensor(a[0]) else None)
        if len(calls) <= n_failures:
            raise torch.OutOfMemoryError("HIP out of memory (injected by the test)")
        return fn(*a, **k)
    return wrapped


def test_offline_model_retries_once_then_reraises():
    import bench
    model, _, _ = bench.build_model("cuda")
    frames = bench.synth_frames(2, 96, 160, 3, "cpu")
    inp = [{"image": [f for f in frames], "dataset_name": "synthetic_burst_val"}]
    ref = model(inp)
    bb = type(model.backbone)
    calls = []
    orig = bb.__call__
    try:
        bb.__call__ = _failing(lambda self, x: orig(self, x), 1, calls)       # one failure: the second attempt succeeds
        out = model(inp)
        assert len(calls) == 2 and out["pred_labels"] == ref["pred_labels"] and out["pred_scores"] == ref["pred_scores"]
        calls.clear()
        bb.__call__ = _failing(lambda self, x: orig(self, x), 5, calls)       # OpenVIS (offline decoder) cannot window: re-raise after 2 attempts
        with pytest.raises(torch.OutOfMemoryError):
            model(inp)
        assert len(calls) == 2
    finally:
        bb.__call__ = orig


def test_online_model_falls_back_to_windows():
    import bench
    model, _, _ = bench.build_model("cuda", model_name="openvis_online")
    model.window_size = 2
    assert model.window_inference is False
    frames = bench.synth_frames(5, 96, 160, 3, "cpu")
    inp = [{"image": [f for f in frames], "dataset_name": "synthetic_burst_val"}]
    ref = model(inp)
    bb = type(model.backbone)
    orig = bb.__call__
    calls = []

    def flaky(self, x):
        calls.append(x.shape[0])
        if x.shape[0] > 2:                                                     # the whole clip does not "fit", a window of 2 frames does
            raise torch.OutOfMemoryError("HIP out of memory (injected by the test)")
        return orig(self, x)
    try:
        bb.__call__ = flaky
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            out = model(inp)
        assert calls == [5, 5, 2, 2, 1]                                        # run, run again, then windows 2 + 2 + 1
        assert any("windows" in str(w.message) for w in wl) and model.window_inference is False and model._fwd.force_windows is False
        assert out["pred_labels"] == ref["pred_labels"]
        assert all(abs(a - b) < 1e-4 for a, b in zip(out["pred_scores"], ref["pred_scores"]))
    finally:
        bb.__call__ = orig
