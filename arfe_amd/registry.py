"""Plugin registries with the contract of the reference's mmdet/models/builder.py:1-48 (which rests on the
absent third-party mmcv.utils.Registry / build_from_cfg): `registry[cfg.pop('type')](**cfg)`, and a LIST of
cfgs becomes an nn.Sequential (builder.py:13-20) — that is how `neck=[FPN, WFPNDualSpatial]` composes."""
import inspect

from torch import nn


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register(self, cls, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        name = cls.__name__
        if not force and name in self._module_dict:
            raise KeyError(f"{name} is already registered in {self._name}")
        self._module_dict[name] = cls

    def register_module(self, cls=None, force=False):
        """Usable both as `@R.register_module` and `@R.register_module()` (the reference uses both forms:
        wfpn_dual_spatial.py:10, multirois_bbox_head.py:11)."""
        if cls is None:
            def deco(c):
                self._register(c, force)
                return c
            return deco
        self._register(cls, force)
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if 'type' not in cfg:
        raise KeyError(f"the cfg dict must contain the key 'type', but got {cfg}")
    if not isinstance(registry, Registry):
        raise TypeError(f"registry must be a Registry, but got {type(registry)}")
    args = dict(cfg)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or a class, but got {type(obj_type)}")
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return obj_cls(**args)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')
# the harness around the hot path: reference mmdet/core/anchor/builder.py:3, core/bbox/builder.py:3-5,
# core/bbox/iou_calculators/builder.py:3
ANCHOR_GENERATORS = Registry('Anchor generator')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
IOU_CALCULATORS = Registry('IoU calculator')


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_roi_extractor(cfg):
    return build(cfg, ROI_EXTRACTORS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:46-48."""
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)
