from .resnet import build_resnet_backbone, ResNet  # noqa: F401
from .swin import D2SwinTransformer, SwinTransformer  # noqa: F401
