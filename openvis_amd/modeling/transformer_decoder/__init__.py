from .video_mask2former_transformer_decoder import (  # noqa: F401
    VideoMultiScaleMaskedTransformerDecoder, build_transformer_decoder)
from .frame_mask2former_transformer_decoder import FrameMultiScaleMaskedTransformerDecoder  # noqa: F401
from .side_adapter_frame_mask2former_transformer_decoder import (  # noqa: F401
    SideAdapterFrameMultiScaleMaskedTransformerDecoder, SideAdapterVideoMultiScaleMaskedTransformerDecoder)
