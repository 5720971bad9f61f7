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
