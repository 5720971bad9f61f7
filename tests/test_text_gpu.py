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
