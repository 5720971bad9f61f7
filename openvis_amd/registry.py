"""Registries with the reference's names (detectron2 META_ARCH_REGISTRY / SEM_SEG_HEADS_REGISTRY /
BACKBONE_REGISTRY; openvis/modeling/transformer_decoder/video_mask2former_transformer_decoder.py:16)."""


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        def deco(o):
            if o.__name__ in self:
                raise KeyError(f"{o.__name__} already registered in {self._name}")
            self[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name):
        if name not in self:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self[name]


META_ARCH_REGISTRY = Registry("META_ARCH")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
BACKBONE_REGISTRY = Registry("BACKBONE")
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
