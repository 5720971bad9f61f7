"""Mirror of openvis/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:21-49 (forward only): the compiled
operator module is imported by its reference name and a missing build fails loudly (no silent torch fallback)."""
try:
    import MultiScaleDeformableAttention as MSDA                     # compiled torch extension at the repo root
except ModuleNotFoundError as e:                                      # func.py:23-29
    raise ModuleNotFoundError(
        "\n\nPlease compile MultiScaleDeformableAttention (the MI355X operator module):\n\n"
        "\t`python -c 'import __graft_entry__ as g; g.build()'`\n") from e


class MSDeformAttnFunction:
    @staticmethod
    def apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        return MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                           attention_weights, im2col_step)
