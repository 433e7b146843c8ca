"""Registries for the drop-in plugin surface.

When mmcv / mmdet / mmdet3d are importable the reference's own registries are used, so
`type='DeepInteractionEncoder'` in the reference configs resolves to these classes.  They
are absent in this image; a local Registry with the same decorator / build surface stands in.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force and self.module_dict[key] is not cls:
                raise KeyError(f'{key} is already registered in {self.name}')
            self.module_dict[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.module_dict[key]

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        cls = self.module_dict[cfg.pop('type')]
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        return cls(**cfg)


def _try(path, name):
    try:
        mod = __import__(path, fromlist=[name])
        return getattr(mod, name)
    except Exception:
        return None


NECKS = _try('mmdet3d.models.builder', 'NECKS') or Registry('neck')
HEADS = _try('mmdet3d.models.builder', 'HEADS') or Registry('head')
TRANSFORMER_LAYER = _try('mmcv.cnn.bricks.registry', 'TRANSFORMER_LAYER') or Registry('transformer layer')
ATTENTION = _try('mmcv.cnn.bricks.registry', 'ATTENTION') or Registry('attention')
BBOX_CODERS = _try('mmdet.core.bbox.builder', 'BBOX_CODERS') or Registry('bbox_coder')


def build_bbox_coder(cfg):
    return BBOX_CODERS.build(cfg)
