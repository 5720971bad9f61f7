"""GPU: the thin eval driver (train_net.py --eval-only) end to end on synthetic clips."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_net_eval_only_writes_ytvis_results(tmp_path):
    from openvis_amd import rle
    out = tmp_path / "results.json"
    cmd = [sys.executable, os.path.join(ROOT, "train_net.py"), "--eval-only", "--synthetic", "2", "--frames", "3", "--output", str(out),
           "MODEL.META_ARCHITECTURE", "OpenVISOnline", "MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME",
           "FrameMultiScaleMaskedTransformerDecoder", "DATASETS.TEST", "['ytvis_2019_val']"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert len(res) == 20 and {x["video_id"] for x in res} == {"synthetic_0", "synthetic_1"}
    for x in res:
        assert 1 <= x["category_id"] <= 40 and 0.0 <= x["score"] <= 1.0 and len(x["segmentations"]) == 3
        seg = x["segmentations"][0]
        assert seg["size"] == [360, 640]
        assert sum(rle.string_to_counts(seg["counts"])) == 360 * 640


def test_train_net_with_videos_in_flight_matches_sequential(tmp_path):
    outs = []
    for streams in ("1", "2"):
        out = tmp_path / f"res_{streams}.json"
        cmd = [sys.executable, os.path.join(ROOT, "train_net.py"), "--eval-only", "--synthetic", "5", "--frames", "2", "--output", str(out),
               "--streams", streams, "DATASETS.TEST", "['ytvis_2019_val']"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.load(open(out)))
    assert outs[0] == outs[1] and len(outs[0]) == 50


def test_train_net_refuses_training():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_net.py"), "--synthetic", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "eval-only" in (r.stderr + r.stdout)


def test_train_net_on_a_directory_of_frames(tmp_path):
    import numpy as np
    from PIL import Image
    from openvis_amd import rle
    rng = np.random.default_rng(0)
    for v in ("vid_a", "vid_b"):
        os.makedirs(tmp_path / "videos" / v)
        for t in range(2):
            Image.fromarray(rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)).save(tmp_path / "videos" / v / f"{t:05d}.png")
    (tmp_path / "classes.txt").write_text("cat\ndog\nzebra\n")
    out = tmp_path / "res.json"
    cmd = [sys.executable, os.path.join(ROOT, "train_net.py"), "--eval-only", "--input", str(tmp_path / "videos"), "--classes",
           str(tmp_path / "classes.txt"), "--output", str(out), "INPUT.MIN_SIZE_TEST", "120"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # default test set of the shipped configs is burst_val (Base.yaml:19): BURST sequences (burst_eval.py:177-240, :160)
    res = json.load(open(out))["sequences"]
    assert {x["seq_name"] for x in res} == {"vid_a", "vid_b"}
    for seq in res:
        assert (seq["height"], seq["width"]) == (200, 300) and seq["annotated_image_paths"] == ["00000.png", "00001.png"]
        assert len(seq["segmentations"]) == 2
        assert all(1 <= c <= 3 for c in seq["track_category_ids"].values())
        for frame in seq["segmentations"]:
            for track_id, a in frame.items():
                assert track_id in seq["track_category_ids"] and a["is_gt"] is False and 0.0 <= a["score"] <= 1.0
                assert sum(rle.string_to_counts(a["rle"])[1::2]) > 20       # the > 20 pixel rule (burst_eval.py:203)
                assert sum(rle.string_to_counts(a["rle"])) == 200 * 300     # masks are at the original frame size
