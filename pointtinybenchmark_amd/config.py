"""Minimal ``Config.fromfile`` for the reference's python config files: ``_base_`` inheritance,
``_delete_`` keys, dotted ``--cfg-options`` overrides, attribute access (mmcv.Config semantics the CPR/P2P
configs rely on: T/configs2/TinyPersonV2/coarsepointv2/coarse_point_refine_base_TinyPersonV2_640.py:8-10,100)."""
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return ConfigDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    if isinstance(o, tuple):
        return tuple(_wrap(v) for v in o)
    return o


def _merge(base, child):
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = copy.deepcopy(v)
    return out


def _load_py(path):
    ns = {'__file__': path}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v)
           and not type(v).__name__ == 'module'}
    base = cfg.pop('_base_', None)
    if base is None:
        return cfg
    merged = {}
    for b in ([base] if isinstance(base, str) else base):
        merged = _merge(merged, _load_py(os.path.normpath(os.path.join(os.path.dirname(path), b))))
    return _merge(merged, cfg)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        return Config(_wrap(_load_py(os.path.abspath(path))))

    def merge_from_dict(self, options):
        """``--cfg-options a.b.c=v`` style overrides (T/tools/train.py:55-57,90-91)."""
        for key, v in options.items():
            d = self
            parts = key.split('.')
            for p in parts[:-1]:
                if p not in d or not isinstance(d[p], dict):
                    d[p] = ConfigDict()
                d = d[p]
            d[parts[-1]] = _wrap(v)
        return self
