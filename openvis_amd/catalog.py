"""Minimal MetadataCatalog (detectron2.data.MetadataCatalog surface used by openvis/openvis.py:43-45).
Dataset registration (openvis/data/datasets/*.py) is out of scope; callers register `thing_classes`."""


class _Metadata:
    def __init__(self, name):
        self.name = name

    def set(self, **kw):
        self.__dict__.update(kw)
        return self

    def __getattr__(self, k):
        raise AttributeError(f"Attribute '{k}' does not exist in the metadata of dataset '{self.name}': "
                             f"register it with MetadataCatalog.get('{self.name}').set({k}=...)")


class _Catalog(dict):
    def get(self, name):
        if name not in self:
            self[name] = _Metadata(name)
        return self[name]


MetadataCatalog = _Catalog()
