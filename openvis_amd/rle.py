"""COCO RLE string codec (host side of SURVEY.md 8f-1).

`counts_to_string` is the published algorithm of pycocotools' rleToString (cocoapi common/maskApi.c): every count after
the third is stored as the difference to the count two places before it, each value is emitted in 5-bit groups, lowest
first, bit 0x20 marks "more groups follow" (decided with sign extension for negative differences) and 48 is added to
make printable ASCII.  pycocotools is not available in this image, so the byte-level format is restated from that
public source ("parity unpinned"); round trips and the run-length semantics are tested."""
import numpy as np


def counts_to_string(counts):
    out = bytearray()
    cnts = [int(c) for c in counts]
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5                                     # arithmetic shift (Python ints are signed)
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return bytes(out)


def string_to_counts(s):
    if isinstance(s, str):
        s = s.encode("ascii")
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def mask_to_counts(mask):
    """numpy reference of the run-length semantics: bool [H,W] -> counts in column-major order, zeros first."""
    flat = np.asarray(mask, dtype=np.uint8).flatten(order="F")
    change = np.flatnonzero(np.diff(np.concatenate([[0], flat])) != 0)
    edges = np.concatenate([[0], change, [flat.size]])
    return np.diff(edges).tolist()


def counts_to_mask(counts, h, w):
    flat = np.zeros(h * w, dtype=np.uint8)
    p, v = 0, 0
    for c in counts:
        flat[p:p + c] = v
        p += c
        v ^= 1
    return flat.reshape((h, w), order="F").astype(bool)


def encode_video_masks(counts, n_runs, n_sel, T, H, W):
    """device results of ops.rle_encode -> list (instance) of list (frame) of {"size": [H, W], "counts": str}: what
    instances_to_coco_json_video builds per frame with mask_util.encode (ytvis_eval.py:283-293)."""
    nr = n_runs.cpu().numpy()
    cap = counts.shape[1]
    if (nr > cap).any():
        raise RuntimeError("RLE buffer too small")
    c = counts[:, :int(nr.max())].cpu().numpy()
    out = []
    for j in range(n_sel):
        out.append([{"size": [H, W], "counts": counts_to_string(c[j * T + t, :nr[j * T + t]]).decode("ascii")} for t in range(T)])
    return out
