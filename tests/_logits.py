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


def check_top10(out_gpu, ref, ref_probs, rows_ref=None, tol=1e-3):
    """Top-10 of the flattened [Q_valid x K] scores (video_maskformer.py:267-278; `topk(sorted=False)`: compared as sets keyed by
    (query, label)).  Exact, tie-aware statement: with d = `tol` the score tolerance,
      * every GPU entry also found in the reference top-10 has |score - ref score| <= d;
      * every reference entry whose score beats the reference's 11th score by more than 2 d must be in the GPU top-10 (the others can
        legitimately swap with the 11th);
      * every GPU entry must score >= the reference's 10th score - 2 d on the reference side.
    Returns (n_common, margin = reference 10th - 11th score).  `rows_ref`: reference row -> query id (rows of valid queries)."""
    p = np.asarray(ref_probs, dtype=np.float64)
    flat = np.sort(p.reshape(-1))[::-1]
    s10, s11 = float(flat[9]), float(flat[10]) if flat.size > 10 else 0.0
    K = p.shape[1]
    qid = (lambda r: rows_ref[r]) if rows_ref is not None else (lambda r: r)
    row_of = {qid(r): r for r in range(p.shape[0])}
    sg = {(q, l): s for q, l, s in zip(out_gpu["pred_queries"], out_gpu["pred_labels"], out_gpu["pred_scores"])}
    sr = {(qid(r), l): s for r, l, s in zip(ref["rows"], ref["pred_labels"], ref["pred_scores"])}
    assert len(sg) == len(sr) == 10
    for k in set(sg) & set(sr):
        assert abs(sg[k] - sr[k]) <= tol, (k, sg[k], sr[k])
    for k, s in sr.items():
        if s > s11 + 2 * tol:
            assert k in sg, ("reference top-10 entry with a clear margin is missing", k, s, s11)
    for (q, l), s in sg.items():
        assert q in row_of and p[row_of[q], l] >= s10 - 2 * tol, ("GPU top-10 entry is not a reference top-10 candidate", q, l, s)
        assert abs(p[row_of[q], l] - s) <= tol, ((q, l), s, p[row_of[q], l])
    return len(set(sg) & set(sr)), s10 - s11


def differing_bits_outside_ambiguous(got, ref, gold):
    """"Masks match bit-exact" as the workload goldens state it (oracle/make_golden_workload.py `ambiguous`): next to the oracle's sign
    bits the fixture holds, for each eps of `ambig_eps`, the bitmap of the pixels whose ORACLE logit lies within eps of zero.
    got / ref: bool arrays [..., h, w].  Returns (n_differing, {eps: number of differing bits OUTSIDE the eps set})."""
    diff = got != ref
    out = {}
    for i, eps in enumerate(gold["ambig_eps"]):
        amb = np.unpackbits(gold[f"ambig_bits_{i}"], axis=-1)[..., : ref.shape[-1]].astype(bool)
        assert amb.shape == ref.shape
        out[float(eps)] = int((diff & ~amb).sum())
    return int(diff.sum()), out
