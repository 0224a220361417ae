"""Configuration surface of the reference (pcdet/config.py:16-86): a global attribute-dict `cfg`, YAML loading with the
one-level `_BASE_CONFIG_` include and `--set KEY VALUE` overrides.  easydict is not a dependency here: AttrDict is a
minimal equivalent (attribute access, nested wrapping, AttributeError for missing keys so copy.deepcopy works)."""
from pathlib import Path

import yaml


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, other=(), **kw):
        for k, v in dict(other, **kw).items():
            self[k] = v


EasyDict = AttrDict


def _flatten(node, prefix):
    """(dotted key, value) pairs in insertion order; a nested section is announced by a (prefix.key, None) header entry first."""
    for key in node:
        val = node[key]
        if isinstance(val, AttrDict):
            yield f"{prefix}.{key}", None
            yield from _flatten(val, f"{prefix}.{key}")
        else:
            yield f"{prefix}.{key}", val


def log_config_to_file(cfg, pre='cfg', logger=None):
    """One line per leaf (`cfg.A.B: value`) and a banner per section, in the format of the reference's training logs
    (pcdet/config.py:7-13)."""
    for dotted, val in _flatten(cfg, pre):
        if val is None:
            logger.info('----------- %s -----------' % dotted.rsplit('.', 1)[1])
        else:
            logger.info('%s: %s' % (dotted, val))


def _parse_scalar(text):
    from ast import literal_eval
    try:
        return literal_eval(text)
    except Exception:
        return text


def _override(section, name, text):
    """Typing rules of `--set` (pcdet/config.py:16-48): the new value must have the type of the old one, except that a section takes
    `k1:v1,k2:v2` (each cast to the existing entry's type) and a list takes `a,b,c` (cast to the type of its first element)."""
    old, new = section[name], _parse_scalar(text)
    if type(new) == type(old):
        section[name] = new
    elif isinstance(old, AttrDict):
        for item in text.split(','):
            sub, sub_text = item.split(':')
            old[sub] = type(old[sub])(sub_text)
    elif isinstance(old, list):          # split the TEXT: `10,20,30` literal-evals to a tuple (on which the reference's value.split fails)
        section[name] = [type(old[0])(piece) for piece in text.split(',')]
    else:
        raise AssertionError('type {} does not match original type {}'.format(type(new), type(old)))


def cfg_from_list(cfg_list, config):
    """`--set A.B value C.D value ...`: every key must already exist."""
    if len(cfg_list) % 2:
        raise AssertionError('--set takes KEY VALUE pairs')
    pairs = zip(cfg_list[::2], cfg_list[1::2])
    for dotted, text in pairs:
        *path, leaf = dotted.split('.')
        section = config
        for name in path:
            if name not in section:
                raise AssertionError('NotFoundKey: %s' % name)
            section = section[name]
        if leaf not in section:
            raise AssertionError('NotFoundKey: %s' % leaf)
        _override(section, leaf, text)


def merge_new_config(config, new_config):
    """Recursive merge of a loaded YAML mapping into `config`; `_BASE_CONFIG_: path` (resolved against the working directory, as in
    the reference, pcdet/config.py:51-67) is loaded into the same level first."""
    base = new_config.get('_BASE_CONFIG_')
    if base is not None:
        with open(base, 'r') as f:
            config.update(AttrDict(yaml.safe_load(f)))
    for key in new_config:
        val = new_config[key]
        if isinstance(val, dict):
            merge_new_config(config.setdefault(key, AttrDict()), val)
        else:
            config[key] = val
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        merge_new_config(config=config, new_config=yaml.safe_load(f))
    return config


cfg = AttrDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
