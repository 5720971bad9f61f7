"""GPU: CLIP text tower + prompt ensemble (A11) vs the reference's CLIP.encode_text / ClipAdapter.encode_text outputs
(tests/golden/clip_text.npz: tokens from the reference tokenizer, features from the reference's modules)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from tests._synth import synth_weights

pytestmark = pytest.mark.gpu


def test_text_tower_and_prompt_ensemble_match_reference():
    from openvis_amd.modeling.clip_adapter.text import ClipText
    g = np.load(os.path.join(GOLDEN, "clip_text.npz"))
    spec = [(k, tuple(s)) for k, s in json.loads(bytes(g["spec"].tolist()).decode())]
    sd = synth_weights(spec, int(g["seeds"][0]), "clip_adapter.clip_model.")
    tt = ClipText.from_state_dict(sd, "clip_adapter.clip_model.", "cuda")
    assert (tt.width, tt.layers, tt.heads) == (64, 2, 1)
    tokens = torch.from_numpy(g["tokens"].astype(np.int64))                  # [14, K, 77]
    e0 = tt.encode_text(tokens[0]).cpu().numpy()
    assert np.abs(e0 - g["encode_text_template0"]).max() < 2e-5
    ens = tt.ensemble(tokens).cpu().numpy()
    assert np.abs(ens - g["ensemble"]).max() < 2e-5
    assert np.abs(np.linalg.norm(ens, axis=-1) - 1).max() < 1e-5


def test_text_tower_real_shape_vs_oracle():
    """A11 at the REAL shape of the shipped configs (CLIP ViT-B/16 text side: width 512, 12 layers, 8 heads, 77 tokens; adapter.py:121-138:
    14 "vild" templates x the 482 burst_val names = 6 748 sequences, 519 596 token rows -- the GEMMs leave the small-problem kernels):
    GPU prompt ensemble for all 482 names against oracle/torch_ref.clip_text_ensemble on 24 of them (the oracle needs ~0.15 s per sequence).
    Random-init weights and synthetic token ids (sot, random word ids, eot = the highest id, zero padding; lengths 4..20 as real prompts):
    the tower's arithmetic does not depend on what the ids mean."""
    from openvis_amd.modeling.clip_adapter.text import ClipText
    from oracle import torch_ref as TR
    width, layers, heads, ctx, vocab, embed, n_tmpl, K = 512, 12, 8, 77, 49408, 512, 14, 482
    spec = [("token_embedding.weight", (vocab, width)), ("positional_embedding", (ctx, width)), ("ln_final.weight", (width,)),
            ("ln_final.bias", (width,)), ("text_projection", (width, embed))]
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        spec += [(p + "attn.in_proj_weight", (3 * width, width)), (p + "attn.in_proj_bias", (3 * width,)), (p + "attn.out_proj.weight", (width, width)),
                 (p + "attn.out_proj.bias", (width,)), (p + "ln_1.weight", (width,)), (p + "ln_1.bias", (width,)), (p + "ln_2.weight", (width,)),
                 (p + "ln_2.bias", (width,)), (p + "mlp.c_fc.weight", (4 * width, width)), (p + "mlp.c_fc.bias", (4 * width,)),
                 (p + "mlp.c_proj.weight", (width, 4 * width)), (p + "mlp.c_proj.bias", (width,))]
    prefix = "clip_adapter.clip_model."
    sd = synth_weights(spec, 77, prefix)
    sd[prefix + "token_embedding.weight"] = sd[prefix + "token_embedding.weight"] * (0.02 * width ** 0.5)     # CLIP init: std 0.02
    sd[prefix + "positional_embedding"] = sd[prefix + "positional_embedding"] * (0.01 * width ** 0.5)
    tt = ClipText.from_state_dict(sd, prefix, "cuda")
    assert (tt.width, tt.layers, tt.heads, tt.context_length) == (width, layers, heads, ctx)
    g = torch.Generator().manual_seed(3)
    tokens = torch.zeros(n_tmpl, K, ctx, dtype=torch.int64)
    lens = torch.randint(4, 21, (n_tmpl, K), generator=g)
    body = torch.randint(1, vocab - 2, (n_tmpl, K, ctx), generator=g)
    for t in range(n_tmpl):
        for k in range(K):
            n = int(lens[t, k])
            tokens[t, k, 0] = vocab - 2                                   # sot
            tokens[t, k, 1:n - 1] = body[t, k, 1:n - 1]
            tokens[t, k, n - 1] = vocab - 1                               # eot: the arg-max position (model.py:487-489)
    ens = tt.ensemble(tokens).cpu()
    assert tuple(ens.shape) == (K, embed) and (ens.norm(dim=-1) - 1).abs().max().item() < 1e-5
    pick = torch.randperm(K, generator=g)[:24]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        ref = TR.clip_text_ensemble(tokens[:, pick], sd, prefix, heads=heads)
    err = (ens[pick] - ref).abs().max().item()
    cos = (ens[pick] * ref).sum(-1)
    print("A11 real shape: max |diff| of the unit ensemble rows %.2e, min cosine with the oracle %.8f" % (err, cos.min().item()))
    assert err < 5e-5 and cos.min().item() > 1 - 1e-6
