"""MSDeformAttn module — mirror of openvis/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:33-125.

Same constructor arguments, parameter names (sampling_offsets / attention_weights / value_proj / output_proj) and
forward signature.  Unlike the reference there is NO silent fallback (its bare `except` at :116-121 degrades to the
slow torch path on any error): every step runs on the HIP kernels or raises."""
import torch

from ..... import ops
from ..functions import MSDeformAttnFunction


class MSDeformAttn:
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.w = {}

    def load_state_dict(self, sd, prefix="", device="cuda"):
        g = lambda k: sd[prefix + k].float().contiguous().to(device)
        for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj"):
            self.w[n + ".weight"], self.w[n + ".bias"] = g(n + ".weight"), g(n + ".bias")
        # one GEMM for [sampling_offsets | attention_weights] (both consume the same query)
        self.w["oa.weight"] = torch.cat([self.w["sampling_offsets.weight"], self.w["attention_weights.weight"]], 0).contiguous()
        self.w["oa.bias"] = torch.cat([self.w["sampling_offsets.bias"], self.w["attention_weights.bias"]], 0).contiguous()
        # ... and, in the encoder under fp16x2, one GEMM for [value_proj | sampling_offsets | attention_weights] (forward_encoder_fused)
        self.w["voa.weight"] = torch.cat([self.w["value_proj.weight"], self.w["oa.weight"]], 0).contiguous()
        self.w["voa.bias"] = torch.cat([self.w["value_proj.bias"], self.w["oa.bias"]], 0).contiguous()
        self._pos_oa = {}
        return self

    def _pos_term(self, pos):
        """pos Woa^T [S, 288]: what the position embedding adds to the offset / weight projection -- frame independent, cached per pos tensor."""
        key = (pos.data_ptr(), tuple(pos.shape))
        hit = self._pos_oa.get(key)
        if hit is None or hit[0] is not pos:
            hit = (pos, ops.gemm_nt(pos, self.w["oa.weight"], None, cw=True))
            self._pos_oa = {key: hit}
        return hit[1]

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """General (drop-in) path: explicit reference points, un-fused op call (ms_deform_attn.py:82-125).  The projections
        and the sampling run on the HIP kernels; the 12-way softmax and `loc = ref + off / (W, H)` are plain torch
        elementwise ops on the device here (this generic entry is not on the model's path: the encoder uses
        forward_encoder_fused, where both are fused into the sampling kernel)."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == Len_in
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = ops.gemm_nt(input_flatten, self.w["value_proj.weight"], self.w["value_proj.bias"], cw=True)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, M, self.d_model // M)
        oa = ops.gemm_nt(query, self.w["oa.weight"], self.w["oa.bias"], cw=True)
        off = oa[..., : M * L * P * 2].reshape(N, Len_q, M, L, P, 2)
        aw = torch.softmax(oa[..., M * L * P * 2:].reshape(N, Len_q, M, L * P), -1).view(N, Len_q, M, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + off / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.contiguous(), aw.contiguous(), self.im2col_step)
        return ops.gemm_nt(out, self.w["output_proj.weight"], self.w["output_proj.bias"], cw=True)

    def forward_encoder_fused(self, query, src, spatial_shapes, level_start_index, residual, shapes_host=None, norm=None, pos=None):
        """Encoder fast path: value_proj + fused [offsets|weights] GEMM + fused softmax/location/sampling kernel +
        output_proj with the residual add fused.  Reference points are the encoder's (msdeformattn.py:155-168).
        norm = (gamma, beta): also the layer's norm1 over the result (ops.gemm_nt_layernorm).
        query = None with pos [S, C] (the frame-independent position embedding, query = src + pos): the three projections run as ONE
        two-output GEMM over src where the kernel allows it (ops.gemm_nt_dual: (src + pos) Woa^T = src Woa^T + pos Woa^T)."""
        value = oa = None
        if query is None:
            C = self.d_model
            both = ops.gemm_nt_dual(src, self.w["voa.weight"], self.w["voa.bias"], self._pos_term(pos), C) if (C % 256 == 0 and pos.dim() == 2) else None
            if both is not None:
                value, oa = both
            else:
                query = ops.add_bcast(src, pos)
        if value is None:
            value = ops.gemm_nt(src, self.w["value_proj.weight"], self.w["value_proj.bias"], cw=True)
            oa = ops.gemm_nt(query, self.w["oa.weight"], self.w["oa.bias"], cw=True)
        samp = ops.msda_encoder_fused(value, oa, spatial_shapes, level_start_index, self.n_heads, self.n_levels,
                                      self.n_points, shapes_host=shapes_host)
        if norm is not None:
            return ops.gemm_nt_layernorm(samp, self.w["output_proj.weight"], self.w["output_proj.bias"], residual, norm[0], norm[1])
        return ops.gemm_nt(samp, self.w["output_proj.weight"], self.w["output_proj.bias"], residual, cw=True)

    __call__ = forward
