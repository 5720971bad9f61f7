"""TEST INFRASTRUCTURE ONLY -- a label space on which the classification half of the path (A10 crop embeddings -> A12 cosine logits ->
per-query mean over frames -> softmax -> A16 top-10; openvis.py:110-147, video_maskformer.py:262-283) can FAIL.

Round 5's review: with random-init CLIP weights and `bench.synth_text` every query scores alike (top-10 margins of 1e-5, one label for all
ten entries), so "top-10 equal or margin small" could not fail.  Two things are needed and both live here / in
`openvis_amd.weights.sharpen_clip_attention`:
  * crop embeddings that differ between queries (peaked attention in the synthetic tower: pairwise cosines 0.58-0.95 instead of 0.9999);
  * text rows built FROM THE ORACLE'S OWN per-query embeddings, so that ten chosen (query, label) pairs with >= 5 distinct labels win with
    un-saturated, graded scores (~0.96 ... 0.58) and the 11th candidate trails the 10th by >= 1e-2 -- far above what the GPU tower's
    rounding moves a score by.  The GPU side gets the same text and must return exactly that set.
Nothing here is imported by the product."""
import numpy as np
import torch


def per_query_mean(embeds, valid):
    """embeds [M, E]: unit crop embeddings in (frame, query) order of `valid` [T, Q] (adapter.py:144-147 order) -> (rows = ids of the queries
    with at least one crop, [R, E] mean embedding of each: what the per-query mean of the cosine logits sees, openvis.py:130-138)."""
    valid = torch.as_tensor(np.asarray(valid)).bool()
    ids = torch.nonzero(valid)                                              # (t, q), lexicographic
    rows = torch.nonzero(valid.any(0))[:, 0]
    e = torch.as_tensor(np.asarray(embeds)).double()
    return rows.tolist(), torch.stack([e[ids[:, 1] == q].mean(0) for q in rows])


def sharp_parts(M, K, seed=0, n_pick=10, n_labels=7, scale=100.0, extra_rows=None):
    """M [R, E] per-query mean embeddings (oracle).  Picks `n_pick` rows far from each other (farthest-point on the cosine), gives them
    `n_labels` distinct labels (n_pick - n_labels labels are shared by two rows: text row = their normalised sum), blends every designed row
    with the common direction c by the beta that maximises the 10th - 11th score margin while the best score stays below 0.97.
    Returns dict(c [E], designed [n_labels, E] (unit, un-blended), labels [n_labels], beta, seed) -- the parts the fixture stores -- and the
    report (top-10 (row, label) pairs, their scores, margin).  extra_rows [X, E]: rows appended to the text for the softmax's denominator only
    (the SideAdapter's non-object embedding: side_adapter.py cal_sim_logits over K + 1 rows, brivis.py:247-252 drops the last column)."""
    g = torch.Generator().manual_seed(int(seed))
    M = torch.as_tensor(np.asarray(M)).double()
    Mn = torch.nn.functional.normalize(M, dim=-1)
    c = torch.nn.functional.normalize(Mn.mean(0), dim=0)
    picked = [int(torch.argmin(Mn @ c))]
    while len(picked) < n_pick:
        mc = (Mn @ Mn[picked].T).max(dim=1).values
        mc[picked] = 2.0
        picked.append(int(torch.argmin(mc)))
    perm = torch.randperm(K, generator=g).tolist()
    best = None
    # candidates: 7 or 10 labels (3 or 0 of them shared by two rows); designed rows from the raw embeddings or from the CENTRED ones (the part of
    # an embedding that is not the common direction separates neighbours better when the per-query means of several frames lie close together)
    for nl in sorted({min(n_labels, K), min(n_pick, K)}):
        labels = perm[:nl]
        lab = [labels[i % nl] for i in range(n_pick)]
        for centred in (False, True):
            src = torch.nn.functional.normalize(Mn - c, dim=-1) if centred else Mn
            designed = torch.stack([torch.nn.functional.normalize(sum(src[p] for p, pl in zip(picked, lab) if pl == l), dim=0) for l in labels])
            for beta in np.linspace(0.02, 1.0, 50):
                parts = dict(c=c, designed=designed, labels=labels, beta=float(beta), seed=int(seed))
                text = text_from_parts(parts, K).double()
                if extra_rows is not None:
                    text = torch.cat([text, torch.as_tensor(np.asarray(extra_rows)).double().reshape(-1, text.shape[1])])
                P = (scale * M @ text.T).softmax(-1)[:, :K]
                fl = P.flatten().sort(descending=True)
                top = sorted((int(i) // K, int(i) % K) for i in fl.indices[:10])
                margin = float(fl.values[9] - fl.values[10])
                if len({l for _, l in top}) >= 5 and float(fl.values[0]) < 0.97 and (best is None or margin > best[1]["margin"]):
                    best = (parts, dict(top=top, scores=fl.values[:11].tolist(), margin=margin, distinct_labels=len({l for _, l in top})))
    if best is None:
        raise RuntimeError("sharp_parts: no blend gives >= 5 distinct labels in the top-10")
    return best


def text_from_parts(parts, K):
    """The [K, E] unit text rows: un-designed classes = normalize(c + 0.5 noise(seed)) (a common direction: scores near the floor for every
    query), designed class l = normalize(beta designed_l + (1 - beta) c).  Deterministic in (parts, K): the golden files store only the parts."""
    c = torch.as_tensor(np.asarray(parts["c"])).double()
    designed = torch.as_tensor(np.asarray(parts["designed"])).double()
    beta = float(np.asarray(parts["beta"]).reshape(-1)[0])
    g = torch.Generator().manual_seed(int(np.asarray(parts["seed"]).reshape(-1)[0]) + 7919)
    noise = torch.randn(K, c.shape[0], generator=g, dtype=torch.float64) / c.shape[0] ** 0.5
    text = torch.nn.functional.normalize(c + 0.5 * noise, dim=-1)
    for row, l in zip(designed, np.asarray(parts["labels"]).reshape(-1).tolist()):
        text[int(l)] = torch.nn.functional.normalize(beta * row + (1 - beta) * c, dim=0)
    return text.float()


def parts_arrays(parts, prefix="sharp_"):
    """-> npz entries"""
    return {prefix + "c": parts["c"].numpy().astype(np.float64), prefix + "designed": parts["designed"].numpy().astype(np.float64),
            prefix + "labels": np.asarray(parts["labels"], np.int64), prefix + "beta": np.asarray([parts["beta"]], np.float64),
            prefix + "seed": np.asarray([parts["seed"]], np.int64)}


def parts_from_arrays(g, prefix="sharp_"):
    return {k: g[prefix + k] for k in ("c", "designed", "labels", "beta", "seed")}
