"""CLIP text tower + prompt ensemble (A11): `CLIP.encode_text` (mask_adapted_clip/model.py:478-491) and
`ClipAdapter.encode_text` / `SideAdapter.encode_text` (adapter.py:121-138; side_adapter.py:211-232) on the gfx950 kernels.

Runs once per vocabulary (14 templates x K names), exact-f32 policy: embedding gather (row-gather kernel) + positional add,
causal masked attention (uint8 mask shared by batch and heads), QuickGELU MLP in the GEMM epilogue, ln_final, gather of
the eot rows, text_projection, L2 normalisation, mean over templates, L2 normalisation."""
import torch

from ... import ops

# openvis/modeling/clip_adapter/text_prompt.py:93-108 ("vild" prompt set of the shipped configs, PROMPT_NAME: "vild")
PREDEFINED_TEMPLATES = {
    "vild": [
        "a photo of a {}.",
        "This is a photo of a {}",
        "There is a {} in the scene",
        "There is the {} in the scene",
        "a photo of a {} in the scene",
        "a photo of a small {}.",
        "a photo of a medium {}.",
        "a photo of a large {}.",
        "This is a photo of a small {}.",
        "This is a photo of a medium {}.",
        "This is a photo of a large {}.",
        "There is a small {} in the scene.",
        "There is a medium {} in the scene.",
        "There is a large {} in the scene.",
    ],
}


class ClipText:
    def __init__(self, width=512, layers=12, heads=8, context_length=77):
        if width // heads != 64:
            raise NotImplementedError("text attention kernel needs head_dim 64")
        self.width, self.layers, self.heads, self.context_length = width, layers, heads, context_length
        self.w = {}

    @classmethod
    def from_state_dict(cls, sd, prefix, device):
        """Build from the text-side keys of a CLIP state dict; None when the checkpoint has no text tower."""
        if prefix + "token_embedding.weight" not in sd:
            return None
        width = sd[prefix + "token_embedding.weight"].shape[1]
        layers = 1 + max(int(k[len(prefix + "transformer.resblocks."):].split(".")[0]) for k in sd
                         if k.startswith(prefix + "transformer.resblocks."))
        return cls(width, layers, width // 64, sd[prefix + "positional_embedding"].shape[0]).load_state_dict(sd, prefix, device)

    def load_state_dict(self, sd, prefix, device):
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        w = self.w
        w["tok"], w["pos"] = g("token_embedding.weight"), g("positional_embedding")
        w["ln_final.w"], w["ln_final.b"] = g("ln_final.weight"), g("ln_final.bias")
        w["proj_t"] = g("text_projection").t().contiguous()
        for i in range(self.layers):
            p = f"transformer.resblocks.{i}."
            for k in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias", "ln_1.weight",
                      "ln_1.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                      "mlp.c_proj.bias"):
                w[f"{i}.{k}"] = g(p + k)
        L = self.context_length
        ld = (L + 3) // 4 * 4
        m = torch.zeros(L, ld, dtype=torch.uint8)
        m[:, :L] = torch.ones(L, L).triu_(1).to(torch.uint8)                   # key j > query i is blocked (model.py:463-469)
        self.causal = m.to(device)
        self.device = device
        return self

    def encode_text(self, tokens):
        """tokens int64 [B, L] -> [B, embed_dim] (un-normalised)."""
        w = self.w
        tokens = tokens.to(self.device)
        B, L = tokens.shape
        C, Hh = self.width, self.heads
        x = torch.empty((B, L, C), dtype=torch.float32, device=self.device)
        ops.batch_index_rows(w["tok"], tokens.to(torch.int32).contiguous(), x, 0, C, L * C, C, C)   # embedding gather
        x = ops.add_bcast(x, w["pos"][:L].contiguous())
        for i in range(self.layers):
            h = ops.layernorm(x, w[f"{i}.ln_1.weight"], w[f"{i}.ln_1.bias"])
            qkv = ops.gemm_nt(h.view(-1, C), w[f"{i}.attn.in_proj_weight"], w[f"{i}.attn.in_proj_bias"])
            att = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B, Hh, L, L, 64, L * 3 * C, 3 * C, L * 3 * C, 3 * C, L * 3 * C,
                                3 * C, mask=self.causal)
            x = ops.gemm_nt(att.view(-1, C), w[f"{i}.attn.out_proj.weight"], w[f"{i}.attn.out_proj.bias"], x.view(-1, C)).view(B, L, C)
            h = ops.layernorm(x, w[f"{i}.ln_2.weight"], w[f"{i}.ln_2.bias"])
            f = ops.gemm_nt(h.view(-1, C), w[f"{i}.mlp.c_fc.weight"], w[f"{i}.mlp.c_fc.bias"], None, ops.ACT_QUICKGELU)
            x = ops.gemm_nt(f, w[f"{i}.mlp.c_proj.weight"], w[f"{i}.mlp.c_proj.bias"], x.view(-1, C)).view(B, L, C)
        x = ops.layernorm(x, w["ln_final.w"], w["ln_final.b"])
        eot = tokens.argmax(dim=-1).to(torch.int32).view(B, 1).contiguous()        # eot token = highest id (model.py:487-489)
        rows = torch.empty((B, 1, C), dtype=torch.float32, device=self.device)
        ops.batch_index_rows(x, eot, rows, L * C, C, C, C, C)
        return ops.gemm_nt(rows.view(B, C), w["proj_t"])

    def ensemble(self, tokens_per_template):
        """int64 [n_templates, K, L] -> unit rows [K, E]: mean over templates of unit embeddings, re-normalised."""
        n, K, L = tokens_per_template.shape
        e = ops.l2norm_rows(self.encode_text(tokens_per_template.reshape(n * K, L)), 1.0).view(n, K, -1)
        return ops.l2norm_rows(ops.mean_over_dim0(e), 1.0)


def encode_nouns(text_tower, tokenizer, templates, nouns):
    """adapter.py:121-138 for a list of (already cleaned) class names -> unit rows [K, E]."""
    toks = torch.stack([tokenizer.tokenize([t.format(n) for n in nouns]) for t in templates])
    return text_tower.ensemble(toks)
