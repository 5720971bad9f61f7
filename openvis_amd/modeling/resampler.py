"""TemporalInstanceResampler — mirror of openvis/modeling/resampler.py:189-316 (eval path).

6 x (temporal self-attention over T per query -> Conv1d k5 / ReLU / Conv1d k3 (replicate padded) + residual ->
LayerNorm -> FFN); prediction heads (mask einsum, attention-bias einsum) are evaluated ONCE on the final state: the
reference runs them (incl. a CLIP back-pass) after every layer and uses only the last at eval (resampler.py:278-296).
State-dict keys as the reference (prefix `resampler.`)."""
import torch

from .. import ops


class TemporalInstanceResampler:
    def __init__(self, hidden_dim=256, feed_dim=2048, nheads=8, nlayers=6, precision="fp16"):
        self.hidden_dim, self.num_heads, self.num_layers = hidden_dim, nheads, nlayers
        self.precision = precision
        self.w, self.h = {}, {}

    def load_state_dict(self, sd, prefix="resampler.", device="cuda"):
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        w = self.w
        for i in range(self.num_layers):
            lp = f"long_aggregate_layers.{i}."
            for k in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                      "norm.weight", "norm.bias"):
                w[f"long{i}.{k}"] = g(lp + k)
            sp = f"short_aggregate_layers.{i}."
            for j in (0, 2):                       # Conv1d weight [Cout, Cin, k] -> GEMM weight [Cout, k*Cin]
                wt = g(f"{sp}{j}.weight")
                w[f"short{i}.{j}.w"] = wt.permute(0, 2, 1).contiguous().view(wt.shape[0], -1)
                w[f"short{i}.{j}.b"] = g(f"{sp}{j}.bias")
            w[f"agg{i}.nw"], w[f"agg{i}.nb"] = g(f"aggregate_norms.{i}.weight"), g(f"aggregate_norms.{i}.bias")
            fp = f"transformer_ffn_layers.{i}."
            for k in ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm.weight", "norm.bias"):
                w[f"ffn{i}.{k}"] = g(fp + k)
        w["decode_norm.w"], w["decode_norm.b"] = g("decode_norm.weight"), g("decode_norm.bias")
        for j in range(3):
            for nm in ("attn_embed", "mask_embed"):
                w[f"{nm}.{j}.w"], w[f"{nm}.{j}.b"] = g(f"{nm}.layers.{j}.weight"), g(f"{nm}.layers.{j}.bias")
        self.h = {k: ops.cast_f16(v) for k, v in w.items() if v.dim() == 2 and v.shape[1] % 8 == 0 and
                  (k.endswith(".w") or k.endswith("weight"))} if self.precision == "fp16" else {}
        return self

    def _mm(self, x, wk, bk=None, residual=None, act=ops.ACT_NONE):
        return ops.gemm_nt(x, self.w[wk], self.w[bk] if bk else None, residual, act, w16=self.h.get(wk), cw=True)

    @staticmethod
    def _temporal_taps(x, k):
        """[T,Q,C] -> [T,Q,k*C]: rows t-(k//2)..t+(k//2) with replicate padding (Conv1d padding='same', replicate)."""
        T = x.shape[0]
        r = k // 2
        idx = (torch.arange(T, device=x.device)[:, None] + torch.arange(-r, r + 1, device=x.device)[None, :]).clamp_(0, T - 1)
        return x[idx].permute(0, 2, 1, 3).contiguous().view(T, x.shape[1], -1)          # gather only

    def temporal(self, frame_embeds):
        """frame_embeds [T,Q,C] (tracker order) -> refined [T,Q,C] (resampler.py:256-289 without the heads)."""
        w = self.w
        T, Q, C = frame_embeds.shape
        H8 = self.num_heads
        D = C // H8
        x = frame_embeds.contiguous()
        for i in range(self.num_layers):
            qkv = self._mm(x, f"long{i}.self_attn.in_proj_weight", f"long{i}.self_attn.in_proj_bias").view(T * Q, 3 * C)
            att = torch.empty((T, Q, C), dtype=torch.float32, device=x.device)
            # attention over time: batch = query (stride 3C), rows = frames (stride Q*3C)
            ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], Q, H8, T, T, D, 3 * C, Q * 3 * C, 3 * C, Q * 3 * C, 3 * C, Q * 3 * C,
                          out=att, o_bs=C, o_ld=Q * C)
            y = self._mm(att, f"long{i}.self_attn.out_proj.weight", f"long{i}.self_attn.out_proj.bias", x)
            x = ops.layernorm(y, w[f"long{i}.norm.weight"], w[f"long{i}.norm.bias"])
            s = self._mm(self._temporal_taps(x, 5), f"short{i}.0.w", f"short{i}.0.b", None, ops.ACT_RELU)
            y = self._mm(self._temporal_taps(s, 3), f"short{i}.2.w", f"short{i}.2.b", x)            # conv + residual
            x = ops.layernorm(y, w[f"agg{i}.nw"], w[f"agg{i}.nb"])
            hdn = self._mm(x, f"ffn{i}.linear1.weight", f"ffn{i}.linear1.bias", None, ops.ACT_RELU)
            y = self._mm(hdn, f"ffn{i}.linear2.weight", f"ffn{i}.linear2.bias", x)
            x = ops.layernorm(y, w[f"ffn{i}.norm.weight"], w[f"ffn{i}.norm.bias"])
        return x

    def prediction_heads(self, x, mask_feats, attn_feats, n_heads_clip):
        """x [t,Q,C]; mask_feats [t,h,w,C]; attn_feats [t,ha,wa,n*C] (NHWC, channel = head*C + c) ->
        (pred_masks [Q,t,h,w], biases [t,n,Q,ha,wa], pred_embeds [t,Q,C]) (resampler.py:304-316 minus the CLIP pass)."""
        w = self.w
        t, Q, C = x.shape
        out = ops.layernorm(x.contiguous(), w["decode_norm.w"], w["decode_norm.b"])
        f16 = self.precision == "fp16"

        def mlp(nm, v):
            v = self._mm(v, f"{nm}.0.w", f"{nm}.0.b", None, ops.ACT_RELU)
            v = self._mm(v, f"{nm}.1.w", f"{nm}.1.b", None, ops.ACT_RELU)
            return self._mm(v, f"{nm}.2.w", f"{nm}.2.b")

        me, ae = mlp("mask_embed", out), mlp("attn_embed", out)
        _, hm, wm, _ = mask_feats.shape
        npx = hm * wm
        mf2 = mask_feats.reshape(-1, C)
        pred_masks = torch.empty((Q, t, hm, wm), dtype=torch.float32, device=x.device)
        ops.gemm_nt_batched(me, mf2, pred_masks, t, Q, npx, C, C, Q * C, C, npx * C, t * npx, npx,
                            b16=ops.cast_f16(mf2) if f16 else None)
        _, ha, wa, nC = attn_feats.shape
        n = n_heads_clip
        npa = ha * wa
        af3 = attn_feats.reshape(t, npa, nC)
        af16 = ops.cast_f16(af3) if f16 else None
        biases = torch.empty((t, n, Q, ha, wa), dtype=torch.float32, device=x.device)
        for i in range(t):
            ops.gemm_nt_batched(ae[i], af3[i], biases[i], n, Q, npa, C, C, 0, nC, C, npa, Q * npa,
                                b16=af16[i] if af16 is not None else None)
        return pred_masks, biases, out
