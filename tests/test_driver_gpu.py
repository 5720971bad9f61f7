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
           "FrameMultiScaleMaskedTransformerDecoder"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert len(res) == 20 and {x["video_id"] for x in res} == {"synthetic_0", "synthetic_1"}
    for x in res:
        assert 1 <= x["category_id"] <= 40 and 0.0 <= x["score"] <= 1.0 and len(x["segmentations"]) == 3
        seg = x["segmentations"][0]
        assert seg["size"] == [360, 640]
        assert sum(rle.string_to_counts(seg["counts"])) == 360 * 640


def test_train_net_refuses_training():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_net.py"), "--synthetic", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "eval-only" in (r.stderr + r.stdout)
