"""Registry + build-from-config-dict: the plugin API the hot path sits behind in the reference
(mmcv.utils.Registry; T/mmdet/models/builder.py:6-44, T/mmdet/core/bbox/builder.py:3-15).
Same class names, same ``dict(type=..., **kwargs)`` contract, so the reference's config files build
unchanged (see config.py)."""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError('%s is already registered in %s' % (key, self.name))
            self._modules[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError('cfg must be a dict with a "type" key, got %r' % (cfg,))
    args = dict(cfg)
    t = args.pop('type')
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError('%s is not in the %s registry' % (t, registry.name))
    return cls(**args)


# all six model registries alias one MODELS registry in the reference (builder.py:6-14)
MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = DETECTORS = ROI_EXTRACTORS = SHARED_HEADS = MODELS
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
MATCH_COST = Registry('match_cost')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg)
                           if (train_cfg is not None or test_cfg is not None) else None)


def build_assigner(cfg, **default_args):
    return BBOX_ASSIGNERS.build(cfg, default_args)


def build_sampler(cfg, **default_args):
    return BBOX_SAMPLERS.build(cfg, default_args)


def build_match_cost(cfg, default_args=None):
    return MATCH_COST.build(cfg, default_args)
