"""Shared by the end-to-end GPU tests: CLIP logit error split by whether a crop's BOX is identical on both sides.

Crops follow masks; a mask that differs from the oracle's in one boundary pixel can move its bounding box, and then the
two towers see different pictures -- that error says nothing about A10/A12.  The north-star bound (1e-3 on the cosine =
1e-1 on the x100 logits) is therefore asserted as a BOUND (max) on every crop whose box is identical, and the crops with a
different box are counted (tests/test_logit_bound_gpu.py feeds identical masks to both sides and bounds the tower alone)."""
import numpy as np


def logit_errors_by_box(st_gpu, st_ref):
    """-> (d_same [n], d_diff [m]): per-crop max |logit diff| for crops valid on both sides, by box equality."""
    vg, vr = st_gpu["valid"], st_ref["valid"].numpy()
    lg, lr = st_gpu["crop_logits"].cpu().numpy(), st_ref["crop_logits"].numpy()
    gb = {(int(c[0]), int(c[1])): c[2:] for c in st_gpu["crops"]}
    rb = st_ref["boxes"].numpy()
    ig = {tuple(x): i for i, x in enumerate(np.argwhere(vg))}
    same, diff = [], []
    for i, (t, q) in enumerate(np.argwhere(vr)):
        k = (int(t), int(q))
        if k not in ig:
            continue
        x0, y0, x1, y1 = gb[k]
        side = max(x1 + 1 - x0, y1 + 1 - y0)
        b = rb[i]
        d = float(np.abs(lg[ig[k]] - lr[i]).max())
        (same if (b[0] == x0 and b[1] == y0 and b[2] == x0 + side and b[3] == y0 + side) else diff).append(d)
    return np.array(same), np.array(diff)
