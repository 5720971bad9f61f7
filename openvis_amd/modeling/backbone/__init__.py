from .resnet import build_resnet_backbone, ResNet  # noqa: F401
