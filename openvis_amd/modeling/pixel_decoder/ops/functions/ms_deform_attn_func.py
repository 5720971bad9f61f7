"""Mirror of openvis/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:32-49 (forward only)."""
from ..... import MultiScaleDeformableAttention as MSDA  # noqa: F401  (fails loudly if the HIP library is missing)
from .....MultiScaleDeformableAttention import ms_deform_attn_forward


class MSDeformAttnFunction:
    @staticmethod
    def apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        return ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                      attention_weights, im2col_step)
