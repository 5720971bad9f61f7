"""MaskFormerHead — mirror of openvis/modeling/mask_former_head.py:18-135 (eval dispatch, :119-135)."""
from ..registry import SEM_SEG_HEADS_REGISTRY
from .pixel_decoder.msdeformattn import build_pixel_decoder
from .transformer_decoder import build_transformer_decoder


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead:
    def __init__(self, input_shape, *, num_classes, pixel_decoder, loss_weight=1.0, ignore_value=-1,
                 transformer_predictor, transformer_in_feature):
        self.in_features = [k for k, _ in sorted(input_shape.items(), key=lambda x: x[1]["stride"])]
        self.ignore_value, self.common_stride, self.loss_weight = ignore_value, 4, loss_weight
        self.pixel_decoder, self.predictor = pixel_decoder, transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes

    @classmethod
    def from_config(cls, cfg, input_shape):
        tif = cfg.MODEL.MASK_FORMER.TRANSFORMER_IN_FEATURE
        if tif != "multi_scale_pixel_decoder":
            raise NotImplementedError(f"TRANSFORMER_IN_FEATURE={tif} is not used by the eval configs of the path")
        return cls({k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
                   ignore_value=cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                   pixel_decoder=build_pixel_decoder(cfg, input_shape), loss_weight=cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
                   transformer_in_feature=tif,
                   transformer_predictor=build_transformer_decoder(cfg, cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, True))

    def load_state_dict(self, sd, prefix="sem_seg_head.", device="cuda"):
        self.pixel_decoder.load_state_dict(sd, prefix + "pixel_decoder.", device)
        self.predictor.load_state_dict(sd, prefix + "predictor.", device)
        return self

    def forward(self, features, mask=None, extra_feats=None, images=None, texts=None):
        mask_features, _, multi_scale_features = self.pixel_decoder.forward_features(features, extra_feats)
        return self.predictor(multi_scale_features, mask_features, mask)

    __call__ = forward
