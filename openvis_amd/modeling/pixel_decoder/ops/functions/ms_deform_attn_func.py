"""Mirror of openvis/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:21-49 (forward only): the compiled
operator module is imported by its reference name and a missing build fails loudly (no silent torch fallback)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), *[os.pardir] * 5))    # the directory that holds openvis_amd/
try:
    try:
        import MultiScaleDeformableAttention as MSDA                 # compiled torch extension, built next to the package
    except ModuleNotFoundError:
        # imported from another working directory: the extension sits beside the package, not necessarily on sys.path
        if _ROOT in sys.path:
            raise
        sys.path.append(_ROOT)
        sys.modules.pop("MultiScaleDeformableAttention", None)
        import MultiScaleDeformableAttention as MSDA
except ModuleNotFoundError as e:                                      # func.py:23-29
    raise ModuleNotFoundError(
        "\n\nPlease compile MultiScaleDeformableAttention (the MI355X operator module):\n\n"
        "\t`python -c 'import __graft_entry__ as g; g.build()'`\n") from e


class MSDeformAttnFunction:
    @staticmethod
    def apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        return MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                           attention_weights, im2col_step)
