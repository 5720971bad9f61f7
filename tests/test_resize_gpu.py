"""GPU: test-time input resize is bit-exact with PIL Image.resize(BILINEAR), the call the reference's
test mapper ends in (augmentation.py:368-373 via detectron2 ResizeShortestEdge)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw,min_size", [((480, 854), 360), ((720, 1280), 360), ((97, 131), 64), ((360, 640), 480), ((50, 400), 32),
                                         ((333, 251), 333)])
def test_resize_matches_pil_bit_exact(hw, min_size):
    from PIL import Image
    from openvis_amd import data
    rng = np.random.default_rng(hw[0] + min_size)
    img = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    img[10:40, 20:90] = 255
    oh, ow = data.shortest_edge_size(hw[0], hw[1], min_size, 1333)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)).transpose(2, 0, 1)
    got = data.resize_frame(torch.from_numpy(img).cuda(), (oh, ow)).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_shortest_edge_size_rule():
    from openvis_amd import data
    assert data.shortest_edge_size(720, 1280, 360) == (360, 640)
    assert data.shortest_edge_size(1280, 720, 360) == (640, 360)
    assert data.shortest_edge_size(480, 854, 360) == (360, 641)          # int(640.5 + 0.5)
    assert data.shortest_edge_size(100, 1000, 360, 1333) == (133, 1333)  # capped by max_size
    assert data.shortest_edge_size(33, 44, 0) == (33, 44)


@pytest.mark.parametrize("hw,min_size", [((480, 854), 360), ((97, 131), 64), ((360, 640), 480), ((333, 251), 333)])
def test_fused_resize_normalise_pad_is_bit_identical_to_resize_then_preprocess(hw, min_size):
    """SURVEY.md 8f-2's sketch: the resize's vertical pass also writes the model's A1 output (normalised, zero padded f32 NHWC4) for the frame
    it produces (csrc/resize.hip: resize_vertical_preprocess_kernel).  Against PIL for the uint8 frames and against resize_frame() +
    VideoMaskFormer.preprocess() for the images: every bit, padding included."""
    from PIL import Image
    import bench
    from openvis_amd import data
    rng = np.random.default_rng(hw[0] * 7 + min_size)
    vids = [rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8) for _ in range(3)]
    model, _, _ = bench.build_model("cuda")
    frames, images, (H, W) = data.resize_and_preprocess(vids, min_size, 1333, model.size_divisibility, model.pixel_mean, model.pixel_std)
    oh, ow = data.shortest_edge_size(hw[0], hw[1], min_size, 1333)
    assert (H, W) == hw and tuple(frames.shape) == (3, 3, oh, ow)
    for i, v in enumerate(vids):
        ref = np.asarray(Image.fromarray(v).resize((ow, oh), Image.BILINEAR)).transpose(2, 0, 1)
        assert np.array_equal(frames[i].cpu().numpy(), ref)
    sep = torch.stack([data.resize_frame(torch.from_numpy(v).cuda(), (oh, ow)) for v in vids])
    ref_img, size, padded = model.preprocess(sep)
    assert size == (oh, ow) and tuple(images.shape) == (3, padded[0], padded[1], 4) and torch.equal(images, ref_img)
    assert float(images[:, oh:].abs().max()) == 0.0 if padded[0] > oh else True


def test_forward_takes_the_fused_a1_output():
    """batched_inputs[0]["images_nhwc4"] (data.resize_and_preprocess) replaces the model's A1 launch: same outputs, and A1 did not run."""
    import bench
    from openvis_amd import data, ops
    rng = np.random.default_rng(5)
    vids = [rng.integers(0, 256, (200, 300, 3), dtype=np.uint8) for _ in range(2)]
    model, _, _ = bench.build_model("cuda")
    frames, images, _ = data.resize_and_preprocess(vids, 96, 1333, model.size_divisibility, model.pixel_mean, model.pixel_std)
    inp = {"image": [f for f in frames], "dataset_name": "synthetic_burst_val"}
    ref = model([dict(inp)])
    calls, real = [], ops.preprocess_u8
    try:
        ops.preprocess_u8 = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        out = model([dict(inp, images_nhwc4=images)])
        n_with = len(calls)
        model([dict(inp)])
    finally:
        ops.preprocess_u8 = real
    assert n_with == 0 and len(calls) == 1
    assert out["pred_labels"] == ref["pred_labels"] and out["pred_scores"] == ref["pred_scores"]
    assert all(torch.equal(a, b) for a, b in zip(out["pred_masks"], ref["pred_masks"]))
