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
        # mask_former_head.py:92-117: the decoder's input width follows TRANSFORMER_IN_FEATURE.  Every config of the reference sets
        # "multi_scale_pixel_decoder" (configs/*/*.yaml:20-21); the other branches exist in the reference's dispatch and are mirrored in
        # forward() below with the failure each of them meets there (all registered decoders are *MultiScale* decoders)
        tif = cfg.MODEL.MASK_FORMER.TRANSFORMER_IN_FEATURE
        if tif not in ("multi_scale_pixel_decoder", "transformer_encoder", "pixel_embedding", "side_adapter") and tif not in input_shape:
            raise KeyError(tif)                                     # input_shape[TRANSFORMER_IN_FEATURE].channels (:101)
        return cls({k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
                   ignore_value=cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                   pixel_decoder=build_pixel_decoder(cfg, input_shape), loss_weight=cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
                   transformer_in_feature=tif,
                   transformer_predictor=build_transformer_decoder(cfg, cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, True))

    def load_state_dict(self, sd, prefix="sem_seg_head.", device="cuda"):
        self.pixel_decoder.load_state_dict(sd, prefix + "pixel_decoder.", device)
        self.predictor.load_state_dict(sd, prefix + "predictor.", device)
        return self

    def forward(self, features, mask=None, extra_feats=None, images=None, texts=None, shard=None):
        """mask_former_head.py:119-135.  "multi_scale_pixel_decoder" is the path; the single-map branches reach decoders that assert three feature
        levels (video decoder:387, frame decoder:98: `assert len(x) == self.num_feature_levels`) and fail there in the reference as well."""
        mask_features, transformer_encoder_features, multi_scale_features = self.pixel_decoder.forward_features(features, extra_feats)
        tif = self.transformer_in_feature
        if tif == "multi_scale_pixel_decoder":
            if shard is not None:                                   # one clip's frames over several GPUs: the offline video decoder's split-KV form
                return self.predictor(multi_scale_features, mask_features, mask, shard=shard)
            return self.predictor(multi_scale_features, mask_features, mask)
        if tif == "side_adapter":                                   # :122: predictor(multi_scale_features, mask_features, images, texts)
            raise TypeError("forward() takes from 3 to 4 positional arguments but 5 were given (no registered decoder takes images and texts: "
                            "mask_former_head.py:122)")
        if tif == "transformer_encoder":
            assert transformer_encoder_features is not None, "Please use the TransformerEncoderPixelDecoder."       # :126-128
            x = transformer_encoder_features
        elif tif == "pixel_embedding":
            x = mask_features                                       # :131
        else:
            x = features[tif]                                       # :133
        assert isinstance(x, (list, tuple)) and len(x) == self.predictor.num_feature_levels, \
            f"TRANSFORMER_IN_FEATURE={tif}: a single map where the multi-scale decoder expects {self.predictor.num_feature_levels} levels"
        return self.predictor(x, mask_features, mask)

    __call__ = forward
