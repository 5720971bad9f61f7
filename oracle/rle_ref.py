"""TEST INFRASTRUCTURE ONLY -- CPU restatement of COCO's run-length mask codec, the checker for openvis_amd/csrc/rle.hip and
openvis_amd/rle.py.

The reference hands masks to pycocotools (`mask_util.encode(np.array(mask[:, :, None], order="F", dtype="uint8"))[0]`,
openvis/data/evals/ytvis_eval.py:282-286; burst_eval.py:196-200).  pycocotools (cocoapi, PythonAPI/pycocotools/_mask.pyx over
common/maskApi.c; un-pinned in the reference's requirements) is not in /root/reference nor in this image, so the three routines
below restate its published algorithm as plain loops:

  rle_encode      <- maskApi.c rleEncode   : column-major scan, counts of alternating runs starting with the ZEROS run
  rle_to_string   <- maskApi.c rleToString : counts[i] - counts[i-2] for i > 2; 5 data bits per character, lowest group first,
                                             bit 0x20 = another group follows (sign-aware), + 48
  rle_from_string <- maskApi.c rleFrString : the inverse

Pinned by hand-derived known answers of rleToString (tests/test_oracle_rle.py: every branch -- multi-group values, negative
differences with and without a continuation); there is no pycocotools here to generate vectors with: "parity unpinned" beyond
those known answers."""


def rle_encode(mask):
    """mask: 2-D array-like of 0/1 [h][w] -> list of run lengths (column-major, first run counts zeros, may be 0)."""
    h = len(mask)
    w = len(mask[0]) if h else 0
    counts, prev, run = [], 0, 0
    for x in range(w):                                   # rleEncode walks M[j] with j = x * h + y
        for y in range(h):
            v = 1 if mask[y][x] else 0
            if v != prev:
                counts.append(run)
                run, prev = 0, v
            run += 1
    counts.append(run)
    return counts


def rle_to_string(counts):
    out = []
    for i in range(len(counts)):
        x = int(counts[i])
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5                                      # arithmetic shift of a signed long
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def rle_from_string(s):
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)                       # sign extension
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(counts, h, w):
    """maskApi.c rleDecode: list of run lengths -> mask [h][w] of 0/1."""
    flat, v = [], 0
    for c in counts:
        flat.extend([v] * c)
        v ^= 1
    assert len(flat) == h * w, "run lengths do not cover the mask"
    return [[flat[x * h + y] for x in range(w)] for y in range(h)]


def rle_encode_np(mask):
    """rle_encode for a numpy bool/0-1 array [h, w], vectorised (the same definition: column-major scan, run lengths starting with the
    zeros run): positions where the scanned value changes -> differences.  tests/test_oracle_rle.py ties it to the loop above; it is what
    lets the 720p GPU test meet the checker instead of the product's own codec."""
    import numpy as np
    flat = np.asarray(mask).astype(np.uint8).T.reshape(-1)           # column-major: j = x * h + y
    if flat.size == 0:
        return [0]
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    return ([0] + counts) if flat[0] else counts
