"""Config system with the reference's semantics (mmcv `Config`, used at tools/train.py:97-99,
tools/test.py:118-129): python config files, `_base_` inheritance (str or list), recursive dict
merge, `_delete_=True` to replace a base dict, `--cfg-options a.b=c` overrides."""
import ast
import copy
import os

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (missing attribute -> AttributeError)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_configdict(v):
    if isinstance(v, dict):
        return ConfigDict({k: _to_configdict(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_to_configdict(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_to_configdict(x) for x in v)
    return v


def _merge_a_into_b(a, b):
    """Recursive merge; a dict in `a` carrying `_delete_=True` replaces b's value wholesale."""
    b = copy.deepcopy(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a dict in the '
                                f'child config but is of type {type(b[k])} in base config; set `{DELETE_KEY}=True`')
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = copy.deepcopy(v)
    return b


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    if not filename.endswith('.py'):
        raise IOError('Only py type configs are supported')
    with open(filename, 'r') as f:
        text = f.read()
    ast.parse(text)  # surface syntax errors with the file name
    ns = {'__file__': filename}
    exec(compile(text, filename, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v)
           and not isinstance(v, type(os))}
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        merged = {}
        for b in base:
            bd, _ = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = merged.keys() & bd.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {dup}')
            merged.update(bd)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg, text


class Config:
    def __init__(self, cfg_dict=None, filename=None, text=''):
        object.__setattr__(self, '_cfg_dict', _to_configdict(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)
        object.__setattr__(self, '_text', text)

    @staticmethod
    def fromfile(filename):
        d, text = _file2dict(filename)
        return Config(d, filename=filename, text=text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _to_configdict(value)

    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """options: {'a.b.c': value} as produced by `--cfg-options`."""
        nested = {}
        for full, v in options.items():
            d = nested
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _to_configdict(_merge_a_into_b(nested, dict(self._cfg_dict))))


def parse_cfg_options(items):
    """['a.b=1', 'c=[1,2]', 'd=foo'] -> dict, values parsed as python literals when possible."""
    out = {}
    for it in items or []:
        k, v = it.split('=', 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out
