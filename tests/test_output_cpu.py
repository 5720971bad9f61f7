"""The deferred hand-off mapping (openvis_amd/output.py) behaves like the reference's plain video_output dict."""


def test_video_output_materialises_on_first_read_and_is_mutable():
    from openvis_amd.output import VideoOutput
    calls = []

    def finish():
        calls.append(1)
        return {"pred_scores": [0.5, 0.25], "pred_masks": ["m0", "m1"]}

    out = VideoOutput({"image_size": (4, 6)}, None, finish)
    assert out.pending and out["image_size"] == (4, 6) and not calls        # fields known at launch time never wait
    out["pred_masks_frames"] = (0, 3)                                        # callers annotate the output (brivis.py)
    assert out.pending and not calls
    assert out["pred_scores"] == [0.5, 0.25] and calls == [1] and not out.pending
    assert dict(out) == {"image_size": (4, 6), "pred_masks_frames": (0, 3), "pred_scores": [0.5, 0.25], "pred_masks": ["m0", "m1"]}
    assert "pred_masks" in out and len(out) == 4 and out.get("missing", 7) == 7
    out.wait()
    assert calls == [1]                                                      # finish() runs once


def test_video_output_contains_and_iteration_wait_first():
    from openvis_amd.output import VideoOutput
    out = VideoOutput({"image_size": (1, 1)}, None, lambda: {"pred_labels": [3]})
    assert "pred_labels" in out and not out.pending
    out2 = VideoOutput({"image_size": (1, 1)}, None, lambda: {"pred_labels": [3]})
    assert sorted(out2) == ["image_size", "pred_labels"]
