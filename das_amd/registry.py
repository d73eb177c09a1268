"""Registry / builder surface of the reference, self-hosted (mmcv and mmdet are not dependencies).

Mirrors `mmdet3d/models/builder.py:6-7,17-99`: `@BACKBONES.register_module()`, `build_backbone`,
`build_neck`, `build_head`, `build_loss`, `build_detector`, `build_model`; `type=` strings resolve
through these registries exactly as in the reference's configs.
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return key in self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        name = name or cls.__name__
        for n in ([name] if isinstance(name, str) else name):
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, got {type(cfg)}')
    if 'type' not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", got {cfg}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    if isinstance(t, str):
        cls = registry.get(t)
        if cls is None:
            raise KeyError(f'{t} is not in the {registry.name} registry')
    elif inspect.isclass(t):
        cls = t
    else:
        raise TypeError(f'type must be a str or class, got {type(t)}')
    return cls(**args)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    return build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg)
