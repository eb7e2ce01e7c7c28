"""Minimal mmcv-style registries so the reference's config dicts resolve unchanged.

Mirrors the contract of SURVEY.md section 8(b): ``@RECOGNIZERS.register_module()`` (libs/models/base.py:8),
``@HEADS`` / ``@OPTIMIZER_BUILDERS`` (libs/models/cil_heads/tsm.py:20,67,189), ``@LOSSES``
(libs/losses/lsc_loss.py:7); lookup by ``dict(type='Name', **kwargs)`` through ``build_model`` /
``build_loss`` / ``build_optimizer`` (libs/cil/cil.py:429,467).  Unknown ``type`` -> ``KeyError``.
"""
from __future__ import annotations

import copy
from typing import Any, Callable, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._module_dict: Dict[str, type] = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key: str):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    def _register(self, cls, name: Optional[str] = None, force: bool = False):
        key = name or cls.__name__
        if not force and key in self._module_dict:
            raise KeyError(f'{key} is already registered in {self._name}')
        self._module_dict[key] = cls

    def register_module(self, name: Optional[str] = None, force: bool = False, module: Optional[type] = None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg: Dict[str, Any], **default_args):
        return build_from_cfg(cfg, self, default_args or None)


def build_from_cfg(cfg: Dict[str, Any], registry: Registry, default_args: Optional[Dict[str, Any]] = None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", but got {cfg}')
    args = copy.deepcopy(dict(cfg))
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif isinstance(obj_type, type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    return obj_cls(**args)


RECOGNIZERS = Registry('recognizer')
BACKBONES = Registry('backbone')
HEADS = Registry('head')
LOSSES = Registry('loss')
OPTIMIZER_BUILDERS = Registry('optimizer builder')
MODELS = RECOGNIZERS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_recognizer(cfg, train_cfg=None, test_cfg=None):
    return RECOGNIZERS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg) if (train_cfg or test_cfg) else RECOGNIZERS.build(cfg)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """``build_model(config.model)`` as called at libs/cil/cil.py:429."""
    return build_recognizer(cfg, train_cfg, test_cfg)
