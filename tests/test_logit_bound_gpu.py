"""GPU: the CLIP cosine-logit BOUND of the north star (1e-3) on the tower alone.  Identical low-res mask logits go to the HIP
ClipAdapter (boxes -> roi_align crops -> ViT-B/16 -> x100 logits, adapter.py:56-147) and to the oracle, so both sides crop
the same boxes with the same soft masks: what is left is the arithmetic of A10/A12.  Measured on MI355X (80 crops x 482
classes, tools/exp_logit_bound.py): fp16 operands + fp16 residual stream (the bench policy) max 9.7e-5, fp16 operands with an f32 stream
5.2e-5, fp32 operands 2.0e-7."""
import pytest

pytestmark = pytest.mark.gpu


def test_clip_logits_within_the_north_star_bound_on_identical_crops():
    from tools.exp_logit_bound import run
    res, ref, _ = run(Q=30, T=2, K=482)
    assert ref.shape[0] >= 50                                  # most of the 60 (frame, query) pairs are valid crops
    assert res["fp16 + fp16 stream"].max().item() <= 3e-4      # bench policy (fp16 operands, fp16 residual stream): measured 1e-4
    assert res["fp16"].max().item() <= 2e-4                    # f32 residual stream: measured 5e-5
    assert res["fp32"].max().item() <= 2e-6
