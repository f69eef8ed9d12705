"""Minimal python-file config loader with `_base_` inheritance and mmengine's dict-merge rules
(child keys override, nested dicts merge recursively, `_delete_=True` replaces), enough to load
config files shaped like the reference's
configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:1-58 and
configs/_base_/yolox_s_8x8_mmyolo.py verbatim."""
import copy
import os
import types


class ConfigDict(dict):
    """dict with attribute access (cfg.model.detector.test_cfg.score_thr)."""

    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError as e:
            raise AttributeError(name) from e
        return v

    def __setattr__(self, name, value):
        self[name] = value

    def get(self, key, default=None):  # noqa: A003
        return dict.get(self, key, default)


def _to_cfgdict(x):
    if isinstance(x, dict):
        return ConfigDict({k: _to_cfgdict(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cfgdict(v) for v in x)
    return x


def _merge(base, child):
    """mmengine Config._merge_a_into_b semantics (a = child, b = base)."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict):
            v = dict(v)
            delete = v.pop('_delete_', False)
            if not delete and isinstance(out.get(k), dict):
                out[k] = _merge(out[k], v)
            else:
                out[k] = copy.deepcopy(v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _load_file(path):
    path = os.path.abspath(path)
    with open(path) as f:
        src = f.read()
    ns = {'__file__': path}
    exec(compile(src, path, 'exec'), ns)  # config files are trusted local python, as in mmengine
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load_file(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        return Config(_to_cfgdict(_load_file(path)))

    def merge_from_dict(self, options):
        """--cfg-options style overrides: {'model.detector.test_cfg.score_thr': 0.1}."""
        for key, v in options.items():
            d = self
            parts = key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _to_cfgdict(v)
