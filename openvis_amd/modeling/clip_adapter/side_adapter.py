"""SideAdapter — mirror of openvis/modeling/clip_adapter/side_adapter.py:81-270 (eval path) on the gfx950 kernels.

front_encode_image: bicubic resize + CLIP normalisation + patchify (one kernel), first `broken_idx` ViT blocks,
merge features after blocks `merge_ids` projected to 256 channels (attn_projs).
post_encode_image: tokens [Q SOS | cls | L patches], additive attention bias built from the decoder's per-head
biases (adaptive max-pool to the token grid), remaining ViT blocks, ln_post, projection, L2 normalisation.
State-dict keys: clip_model.* (OpenAI CLIP), attn_projs.{i}.{weight,bias}, bg_embed."""
import torch

from ... import ops
from .adapter import ClipVisual, PIXEL_MEAN, PIXEL_STD, _CLIP_ARCH


class SideAdapter:
    def __init__(self, clip_model_name="ViT-B/16", out_dims=256, broken_idx=9, merge_ids=(3, 6, 9), num_queries=100,
                 text_templates="vild", arch=None, precision="fp16"):
        self.arch = dict(arch or _CLIP_ARCH[clip_model_name])
        self.precision = precision
        self.visual = ClipVisual(**self.arch, precision=precision)
        self.input_resolution = self.arch["resolution"]
        self.vis_width, self.embed_dims = self.arch["width"], self.arch["embed_dim"]
        self.num_heads = self.vis_width // 64
        self.grid_size = self.arch["resolution"] // self.arch["patch"]
        self.num_layers = self.arch["layers"]
        self.broken_idx, self.merge_ids = broken_idx, list(merge_ids)
        self.out_dims, self.sos_token_num = out_dims, num_queries
        self.templates = text_templates
        self.text_cache = {}
        self.w = {}

    def load_state_dict(self, sd, prefix="clip_adapter.", device="cuda"):
        self.device = device
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        from .text import ClipText
        self.visual.load_state_dict(sd, prefix + "clip_model.visual.", device)
        self.text_tower = ClipText.from_state_dict(sd, prefix + "clip_model.", device)     # None: vision-only checkpoint
        self.tokenizer = None
        for i in range(len(self.merge_ids)):
            wt = g(f"attn_projs.{i}.weight")
            self.w[f"attn_projs.{i}.w"] = wt.view(wt.shape[0], wt.shape[1]).contiguous()
            self.w[f"attn_projs.{i}.b"] = g(f"attn_projs.{i}.bias")
        self.w["bg_embed"] = g("bg_embed")
        self.logit_scale_exp = float(sd[prefix + "clip_model.logit_scale"].float().exp())
        return self

    # ---- text side (side_adapter.py:211-232) -----------------------------------------------------
    def set_text_features(self, noun_list, feats):
        feats = feats.float().to(self.device)
        self.text_cache.update(dict(zip([self._clean(w) for w in noun_list], feats)))

    @staticmethod
    def _clean(word):
        return word.replace("(", "").replace(")", "").replace("_", " ")       # side_adapter.py:214

    def encode_text(self, x, w_bg=True):
        x = [self._clean(w) for w in x]
        missing = [w for w in dict.fromkeys(x) if w not in self.text_cache]
        if missing:
            if getattr(self, "text_tower", None) is None:
                raise NotImplementedError(f"no cached text embedding for {missing[:3]}... and the checkpoint has no CLIP "
                                          "text tower: call set_text_features first")
            from .text import PREDEFINED_TEMPLATES, encode_nouns
            if self.tokenizer is None:
                from ...simple_tokenizer import SimpleTokenizer
                self.tokenizer = SimpleTokenizer()
            templates = PREDEFINED_TEMPLATES[self.templates] if isinstance(self.templates, str) else self.templates
            self.text_cache.update(dict(zip(missing, encode_nouns(self.text_tower, self.tokenizer, templates, missing))))
        cat = torch.stack([self.text_cache[w] for w in x])
        if w_bg:
            bg = ops.l2norm_rows(self.w["bg_embed"].view(1, -1).contiguous(), 1.0)
            cat = torch.cat([cat, bg], dim=0)
        return cat.contiguous()

    def cal_sim_logits(self, text_feats, image_feats):
        """exp(logit_scale) * image_feats @ text_feats.T (side_adapter.py:234-235); image_feats [T,Q,E] unit rows."""
        T, Q, E = image_feats.shape
        scaled = ops.l2norm_rows(image_feats.view(T * Q, E), self.logit_scale_exp)     # rows are unit: re-scaling only
        return ops.gemm_nt(scaled, text_feats).view(T, Q, -1)

    # ---- image side ------------------------------------------------------------------------------
    def front_encode_image(self, frames_u8, padded_hw):
        """frames uint8 [T,3,H,W] (raw); returns (mg_feats: 3 x [T,g,g,out_dims] NHWC, tokens f32 [T, 1+g*g, C])."""
        T = frames_u8.shape[0]
        Hp, Wp = padded_hw
        A = ops.san_front_patches(frames_u8, Hp, Wp, self.input_resolution, self.arch["patch"], PIXEL_MEAN, PIXEL_STD,
                                  out_f16=(self.precision == "fp16"))
        x = self.visual.embed(A, T)                                           # pos-embed grid == token grid: no resize
        g = self.grid_size
        C = self.vis_width
        mg, done = [], 0
        for j, mid in enumerate(self.merge_ids):
            if mid > 0:
                x = self.visual.run_blocks(x, done, mid)
                done = mid
            pix = x[:, 1:, :].contiguous().view(T * g * g, C)
            mg.append(ops.gemm_nt(pix, self.w[f"attn_projs.{j}.w"], self.w[f"attn_projs.{j}.b"], cw=True).view(T, g, g, -1))
        if done < self.broken_idx:
            x = self.visual.run_blocks(x, done, self.broken_idx)
        return mg, x

    def post_encode_image(self, tokens, attn_bias):
        """tokens f32 [T, 1+L, C] (after block `broken_idx`); attn_bias [T, n, Q, h, w] -> sos features [T,Q,E] (unit rows)."""
        T, L1, C = tokens.shape
        Q, g = self.sos_token_num, self.grid_size
        n = attn_bias.shape[1]
        pooled = ops.adaptive_maxpool2d(attn_bias.contiguous(), g, g).view(T, n, Q, g * g)
        if n == 1:
            pooled = pooled.expand(T, self.num_heads, Q, g * g).contiguous()
        bias = ops.san_attn_bias(pooled, Q, g * g)                             # [T, heads, Q+1+L, ld]
        sos = tokens[:, :1, :].expand(T, Q, C)
        x = torch.cat([sos, tokens], dim=1).contiguous()                      # [T, Q+1+L, C] (copy only)
        x = self.visual.run_blocks(x, self.broken_idx, self.num_layers, attn_bias=bias)
        feat = self.visual.head(x[:, :Q, :].contiguous().view(T * Q, C))
        return ops.l2norm_rows(feat, 1.0).view(T, Q, -1)
