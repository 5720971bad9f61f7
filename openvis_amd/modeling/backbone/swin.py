"""Swin Transformer backbone (`D2SwinTransformer`, openvis/modeling/backbone/swin.py:743-769) on the gfx950 kernels.

The token map stays `[B,H,W,C]` (channel-last = the layout the pixel decoder wants, so the reference's final
`permute(0,3,1,2)` disappears).  Per block (swin.py:224-284):
  LayerNorm (fp16 result under the autocast policy) -> [pad + roll + window_partition: ONE kernel] -> qkv GEMM -> flash window attention (relative-position
  bias as an additive f32 table, shifted-window mask as a uint8 table, both built once per stage/geometry)
  -> proj GEMM -> [window_reverse + roll + crop + residual: ONE kernel] -> LayerNorm -> fc1 + exact GELU (epilogue)
  -> fc2 + residual (epilogue).
Zero-padded tokens go through the qkv GEMM like the reference's (their q/k/v are the qkv bias), so padded keys take
part in the softmax exactly as in swin.py:241-262.  The shifted-window mask is -100 in the reference; here those keys
are excluded, which differs by exp(-100) ~ 4e-44 of the softmax mass (below f32 resolution).
State-dict keys are the reference's (`patch_embed.proj`, `layers.{i}.blocks.{j}.{norm1,attn.qkv,attn.proj,
attn.relative_position_bias_table,norm2,mlp.fc1,mlp.fc2}`, `layers.{i}.downsample.{norm,reduction}`, `norm{i}`)."""
import torch

from ...config import backbone_precision as _backbone_precision
from ... import ops
from ...registry import BACKBONE_REGISTRY


class SwinTransformer:
    size_divisibility = 32

    def __init__(self, patch_size=4, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, ape=False, patch_norm=True, out_features=("res2", "res3", "res4", "res5"),
                 precision="fp16"):
        self.ape = bool(ape)                         # MODEL.SWIN.APE (swin.py:567-578): no shipped config sets it
        if patch_size != 4:
            raise NotImplementedError("MODEL.SWIN.PATCH_SIZE != 4")
        self.patch_size, self.embed_dim, self.depths, self.num_heads = patch_size, embed_dim, tuple(depths), tuple(num_heads)
        self.window_size, self.mlp_ratio, self.qkv_bias, self.qk_scale = window_size, mlp_ratio, qkv_bias, qk_scale
        self.patch_norm = patch_norm
        self.out_features = tuple(out_features)
        self.num_features = [int(embed_dim * 2 ** i) for i in range(len(depths))]
        for c, h in zip(self.num_features, self.num_heads):
            if c // h != 32 or c % h:
                raise NotImplementedError(f"window attention kernel needs head_dim 32 (got {c}/{h})")
            if qk_scale is not None and abs(qk_scale - 32 ** -0.5) > 1e-12:
                raise NotImplementedError("MODEL.SWIN.QK_SCALE other than head_dim ** -0.5")
        self.precision = precision
        self.w, self.w16 = {}, {}
        self._tables = {}

    def output_shape(self):
        return {f"res{i + 2}": dict(channels=self.num_features[i], stride=4 * 2 ** i) for i in range(len(self.depths))
                if f"res{i + 2}" in self.out_features}

    # ---- weights --------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, prefix="backbone.", device="cuda"):
        g = lambda k: sd[prefix + k].float().to(device).contiguous()
        w = self.w
        pw = sd[prefix + "patch_embed.proj.weight"].float().permute(0, 2, 3, 1)             # [E,4,4,3] -> pad Cin to 4
        w["patch_embed.proj.weight"] = torch.nn.functional.pad(pw, (0, 1)).contiguous().to(device)
        w["patch_embed.proj.bias"] = g("patch_embed.proj.bias")
        names = []
        if self.patch_norm:
            names += ["patch_embed.norm.weight", "patch_embed.norm.bias"]
        for i, depth in enumerate(self.depths):
            for j in range(depth):
                p = f"layers.{i}.blocks.{j}."
                names += [p + n for n in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.proj.weight", "attn.proj.bias",
                                          "attn.relative_position_bias_table", "norm2.weight", "norm2.bias", "mlp.fc1.weight",
                                          "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")]
                if self.qkv_bias:
                    names.append(p + "attn.qkv.bias")
            if i < len(self.depths) - 1:
                names += [f"layers.{i}.downsample.{n}" for n in ("norm.weight", "norm.bias", "reduction.weight")]
            if f"res{i + 2}" in self.out_features:
                names += [f"norm{i}.weight", f"norm{i}.bias"]
        for n in names:
            w[n] = g(n)
        # absolute position embedding [1,E,h0,w0] (pretrain grid): kept on the host; the table for an input size is a
        # weight-derived constant like the relative position bias tables (built once per size in _ape_table)
        self._ape = sd[prefix + "absolute_pos_embed"].float() if self.ape else None
        self.w16 = {k: ops.cast_f16(v) for k, v in w.items()
                    if self.precision == "fp16" and k.endswith(".weight") and v.dim() == 2 and "norm" not in k}
        self._tables = {}
        return self

    def _lin(self, x, name, act=ops.ACT_NONE, residual=None, bias=True, out_f16=False):
        """fp32 policy: exact-f32 GEMM.  fp16 policy (= the reference under autocast): fp16 activations x fp16 weights on the
        persistent fp16 MFMA kernel, f32 accumulation, f32 bias / residual, f32 or fp16 result."""
        b = self.w.get(name + ".bias") if bias else None
        if x.dtype == torch.float16:
            return ops.gemm_nt_f16(x, self.w16[name + ".weight"], b, residual, act, out_f16=out_f16)
        return ops.gemm_nt(x, self.w[name + ".weight"], b, residual, act, cw=True)

    def _ape_table(self, H, W, device):
        """absolute_pos_embed bicubically interpolated to the patch grid (swin.py:656-661) as [H,W,E] NHWC, cached per size."""
        k = ("ape", H, W)
        if k not in self._tables:
            t = torch.nn.functional.interpolate(self._ape, size=(H, W), mode="bicubic")        # constant preparation, once per size
            self._tables[k] = t[0].permute(1, 2, 0).contiguous().to(device)
        return self._tables[k]

    def _stage_tables(self, i, j, B, H, W, device):
        """(relative position bias [heads,N,ld] of block (i,j), shift mask [B*nW,N,ld] or None) -- cached per geometry."""
        ws, heads = self.window_size, self.num_heads[i]
        N = ws * ws
        ld = (N + 3) // 4 * 4
        kb = ("bias", i, j)
        if kb not in self._tables:
            self._tables[kb] = ops.swin_relpos_bias(self.w[f"layers.{i}.blocks.{j}.attn.relative_position_bias_table"], heads, ws, ld)
        km = ("mask", i, B, H, W)
        if km not in self._tables:
            m = ops.swin_shift_mask(H, W, ws, ws // 2, ld, device)                             # [nW,N,ld]
            # the f32 flash kernel indexes the mask by batch (b*nW + w); the window kernel by (b*nW + w) mod nW
            self._tables[km] = m if self.precision == "fp16" else m.repeat(B, 1, 1).contiguous()
        return self._tables[kb], self._tables[km], ld

    # ---- forward --------------------------------------------------------------------------------------------------
    def _block(self, x, i, j):
        B, H, W, C = x.shape
        ws, heads = self.window_size, self.num_heads[i]
        shift = 0 if j % 2 == 0 else ws // 2
        p = f"layers.{i}.blocks.{j}."
        bias, mask, ld = self._stage_tables(i, j, B, H, W, x.device)
        f16 = self.precision == "fp16"
        h = ops.layernorm(x, self.w[p + "norm1.weight"], self.w[p + "norm1.bias"], out_f16=f16)
        win = ops.swin_window_partition(h, ws, shift)                                            # [B*nW, N, C]
        nwin, N = win.shape[0], ws * ws
        if f16:
            qkv = self._lin(win.view(-1, C), p + "attn.qkv", bias=self.qkv_bias, out_f16=True).view(nwin, N, 3 * C)
            a = ops.swin_window_attention_f16(qkv, bias, mask if shift > 0 else None, heads)
        else:
            qkv = self._lin(win.view(-1, C), p + "attn.qkv", bias=self.qkv_bias).view(nwin, N, 3 * C)
            a = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], nwin, heads, N, N, 32, N * 3 * C, 3 * C,
                              N * 3 * C, 3 * C, N * 3 * C, 3 * C, mask=mask if shift > 0 else None, mask_per_batch=True,
                              bias=bias, bias_strides=(0, N * ld))
        a = self._lin(a.view(-1, C), p + "attn.proj").view(nwin, N, C)
        x = ops.swin_window_merge_add(a, x, ws, shift)                                            # shortcut + attention
        h = ops.layernorm(x, self.w[p + "norm2.weight"], self.w[p + "norm2.bias"], out_f16=f16)
        h = self._lin(h.view(-1, C), p + "mlp.fc1", act=ops.ACT_GELU, out_f16=f16)
        return self._lin(h, p + "mlp.fc2", residual=x.view(-1, C)).view(B, H, W, C)

    def forward(self, x):
        """x: f32 [T,Hp,Wp,4] (normalised, channel 3 zero; Hp, Wp multiples of 32) -> {res2..res5} NHWC."""
        w = self.w
        x = ops.conv2d_nhwc(x, w["patch_embed.proj.weight"], self.patch_size, 0, w["patch_embed.proj.bias"], cw=True)
        if self.patch_norm:
            x = ops.layernorm(x, w["patch_embed.norm.weight"], w["patch_embed.norm.bias"])
        if self.ape:
            x = ops.add_bcast(x, self._ape_table(x.shape[1], x.shape[2], x.device))               # x + ape (swin.py:713)
        feats = {}
        for i, depth in enumerate(self.depths):
            for j in range(depth):
                x = self._block(x, i, j)
            if f"res{i + 2}" in self.out_features:
                feats[f"res{i + 2}"] = ops.layernorm(x, w[f"norm{i}.weight"], w[f"norm{i}.bias"])
            if i < len(self.depths) - 1:
                B, H, W, C = x.shape
                g = ops.swin_patch_merge_gather(x)                                               # [B,H/2,W/2,4C]
                g = ops.layernorm(g, w[f"layers.{i}.downsample.norm.weight"], w[f"layers.{i}.downsample.norm.bias"],
                                  out_f16=self.precision == "fp16")
                x = self._lin(g.view(-1, 4 * C), f"layers.{i}.downsample.reduction", bias=False).view(B, g.shape[1], g.shape[2], 2 * C)
        return feats

    __call__ = forward


@BACKBONE_REGISTRY.register()
def D2SwinTransformer(cfg, input_shape=None):
    s = cfg.MODEL.SWIN
    return SwinTransformer(s.PATCH_SIZE, s.EMBED_DIM, s.DEPTHS, s.NUM_HEADS, s.WINDOW_SIZE, s.MLP_RATIO, s.QKV_BIAS, s.QK_SCALE,
                           s.APE, s.PATCH_NORM, s.OUT_FEATURES,
                           precision=_backbone_precision(cfg))
