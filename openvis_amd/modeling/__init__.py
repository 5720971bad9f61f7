from . import backbone, pixel_decoder, transformer_decoder, clip_adapter  # noqa: F401
from . import mask_former_head, video_maskformer, minvis  # noqa: F401  (registers MaskFormerHead / VideoMaskFormer / MinVIS)
