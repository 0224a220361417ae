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


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key, val in cfg.items():
        if isinstance(cfg[key], AttrDict):
            logger.info('----------- %s -----------' % key)
            log_config_to_file(cfg[key], pre=pre + '.' + key, logger=logger)
            continue
        logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """`--set A.B value ...` overrides with the reference's typing rules (config.py:16-48)."""
    from ast import literal_eval
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split('.')
        d = config
        for subkey in key_list[:-1]:
            assert subkey in d, 'NotFoundKey: %s' % subkey
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d, 'NotFoundKey: %s' % subkey
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        if type(value) != type(d[subkey]) and isinstance(d[subkey], AttrDict):
            for src in value.split(','):
                cur_key, cur_val = src.split(':')
                d[subkey][cur_key] = type(d[subkey][cur_key])(cur_val)
        elif type(value) != type(d[subkey]) and isinstance(d[subkey], list):
            val_list = value.split(',')
            d[subkey] = [type(d[subkey][0])(x) for x in val_list]
        else:
            assert type(value) == type(d[subkey]), 'type {} does not match original type {}'.format(type(value), type(d[subkey]))
            d[subkey] = value


def merge_new_config(config, new_config):
    if '_BASE_CONFIG_' in new_config:
        with open(new_config['_BASE_CONFIG_'], 'r') as f:       # resolved relative to the CWD, like the reference
            config.update(AttrDict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = AttrDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config=config, new_config=new_config)
    return config


cfg = AttrDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
