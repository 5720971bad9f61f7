"""CPU: the round-6 fixture machinery -- the separated label space (oracle/fixtures.py), the golden files built with it, and the
autocast-arithmetic ResNet of the oracle (oracle/torch_ref.py: resnet50_autocast) -- checked against themselves and against the files."""
import os

import numpy as np
import torch

from oracle import fixtures as FX
from oracle import torch_ref as TR

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_sharp_parts_separate_ten_winners_and_round_trip_through_arrays():
    g = torch.Generator().manual_seed(5)
    base = torch.nn.functional.normalize(torch.randn(1, 64, generator=g), dim=-1)
    M = torch.nn.functional.normalize(base + 0.6 * torch.randn(40, 64, generator=g) / 8, dim=-1)          # 40 rows around one direction
    parts, rep = FX.sharp_parts(M, 30)
    assert rep["margin"] >= 1e-2 and rep["distinct_labels"] >= 5 and max(rep["scores"]) < 0.97 and len(rep["top"]) == 10
    text = FX.text_from_parts(parts, 30)
    assert text.shape == (30, 64) and torch.allclose(text.norm(dim=-1), torch.ones(30), atol=1e-6)
    again = FX.text_from_parts(FX.parts_from_arrays(FX.parts_arrays(parts)), 30)                            # what a golden file stores
    assert torch.equal(text, again)
    P = (100.0 * M.double() @ text.double().T).softmax(-1)
    fl = P.flatten().sort(descending=True)
    assert sorted((int(i) // 30, int(i) % 30) for i in fl.indices[:10]) == rep["top"]


def test_c2_sharp_classes_golden_is_consistent_with_its_own_embeddings():
    """tests/golden/c2_sharp_classes.npz: the stored class probabilities / top-10 follow from the stored crop embeddings and the text rebuilt
    from the stored parts through the oracle's own aggregation (openvis.py:126-142) -- the file cannot drift from the functions that read it."""
    g = np.load(os.path.join(GOLDEN, "c2_sharp_classes.npz"))
    K = 482
    text = FX.text_from_parts(FX.parts_from_arrays(g), K)
    E, valid = torch.from_numpy(g["crop_embeds"]), torch.from_numpy(g["valid"].astype(bool))
    logits = 100.0 * E @ text.T
    probs, _, _ = TR.aggregate_crop_logits(logits, valid, torch.zeros((valid.shape[1], valid.shape[0], 1, 1)))
    assert np.abs(probs.numpy() - g["probs"]).max() < 1e-6
    fl = probs.flatten().sort(descending=True)
    assert sorted((int(i) // K, int(i) % K) for i in fl.indices[:10]) == sorted(zip(g["top_rows"].tolist(), g["top_labels"].tolist()))
    assert float(fl.values[9] - fl.values[10]) >= 1e-2 and abs(float(fl.values[9] - fl.values[10]) - float(g["margin"][0])) < 1e-4
    assert len(set(g["top_labels"].tolist())) >= 5 and g["rows"].tolist() == torch.nonzero(valid.any(0))[:, 0].tolist()


def test_c2_autocast_golden_counts_match_its_bits():
    a = np.load(os.path.join(GOLDEN, "c2_autocast_backbone.npz"))
    f = np.load(os.path.join(GOLDEN, "c2_openvis_720p_5f.npz"))
    w = int(a["mask_shape"][-1])
    auto = np.unpackbits(a["mask_bits"], axis=-1)[..., :w].astype(bool)
    ref = np.unpackbits(f["mask_bits"], axis=-1)[..., :w].astype(bool)
    diff = auto != ref
    assert int(diff.sum()) == int(a["n_diff_vs_f32"][0]) == 35575
    for i in range(3):
        amb = np.unpackbits(f[f"ambig_bits_{i}"], axis=-1)[..., :w].astype(bool)
        assert int((diff & ~amb).sum()) == int(a["outside_vs_f32"][i])


def test_resnet50_autocast_keeps_every_tensor_in_fp16_and_stays_near_the_f32_backbone():
    """oracle resnet50_autocast (the reference's GPU arithmetic under autocast, train_net.py:241): the outputs hold fp16 values, differ from the
    f32 backbone by fp16-sized amounts, and DO differ (the mode is not a no-op)."""
    from openvis_amd import weights
    spec = [(k, s) for k, s in weights.openvis_spec("r50", None, 100) if k.startswith("backbone.")]
    sd = weights.random_init(spec, seed=3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 3, 64, 96, generator=g)
    with torch.no_grad():
        f32, f16 = TR.resnet50(x, sd), TR.resnet50_autocast(x, sd)
    for k in ("res2", "res3", "res4", "res5"):
        assert f16[k].shape == f32[k].shape and torch.equal(f16[k], f16[k].half().float())
        rel = (f16[k] - f32[k]).abs().max().item() / f32[k].abs().max().item()
        assert 1e-5 < rel < 3e-2, (k, rel)
