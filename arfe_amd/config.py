"""Python-file configs with `_base_` inheritance and `_delete_` keys — the behaviour of the absent third-party
mmcv.Config that the reference's tools/train.py:68 relies on — enough to load configs/faster_rcnn/*,
configs/cascade_rcnn/* and configs/mytrain/* unchanged and hand their `neck` / `roi_head` dicts to the registries."""
import ast
import os
import warnings


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _to_cfgdict(o):
    if isinstance(o, dict):
        return ConfigDict({k: _to_cfgdict(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_to_cfgdict(v) for v in o]
    if isinstance(o, tuple):
        return tuple(_to_cfgdict(v) for v in o)
    return o


def _merge(a, b):
    """b into a copy of base a; a dict in b carrying `_delete_=True` replaces instead of merging."""
    out = dict(a)
    for k, v in b.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = v
    return out


def _exec_file(path):
    with open(path, 'r') as f:
        text = f.read()
    try:
        tree = ast.parse(text, filename=path)
    except SyntaxError as e:
        # the reference's flagship AR config carries a stray back-tick (configs/mytrain/
        # faster_rcnn_r50_drfpn_drrff+fac_1x_coco.py:21); tolerate exactly that and say so
        cleaned = text.replace('`', '')
        try:
            tree = ast.parse(cleaned, filename=path)
        except SyntaxError:
            raise e
        warnings.warn(f"{path}: removed stray '`' characters to parse the config (line {e.lineno})")
    scope = {'__file__': path}
    exec(compile(tree, path, 'exec'), scope)
    return {k: v for k, v in scope.items() if not k.startswith('__') and not callable(v) and
            not isinstance(v, type(os))}


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _to_cfgdict(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        cfg = _exec_file(filename)
        base = cfg.pop('_base_', None)
        if base is not None:
            base = base if isinstance(base, list) else [base]
            merged = {}
            for b in base:
                bd = Config._file2dict(os.path.join(os.path.dirname(filename), b))
                dup = set(merged) & set(bd)
                if dup:
                    raise KeyError(f"duplicate keys in _base_ files of {filename}: {sorted(dup)}")
                merged.update(bd)
            cfg = _merge(merged, cfg)
        return cfg

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename=filename)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def merge_from_dict(self, options):
        """`--options k1.k2=v` of tools/train.py:46-47,69-70."""
        d = {}
        for full, v in options.items():
            cur = d
            keys = full.split('.')
            for k in keys[:-1]:
                cur = cur.setdefault(k, {})
            cur[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _to_cfgdict(_merge(self._cfg_dict, d)))
