"""mmcv-compatible Registry / build_from_cfg / ConfigDict / Config.

The reference resolves every class on the hot path through mmcv registries named in the
python-dict configs (mmdet/models/builder.py:7-15, mmdet/core/bbox/builder.py:4-16,
mmdet/core/anchor/builder.py:6-12).  mmcv is un-vendored and not installable here, so the
same surface is provided natively; only the behaviour the shipped configs use is kept.
"""
import os


class ConfigDict(dict):
    """dict with attribute access (mmcv.ConfigDict stand-in)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def wrap(obj):
        if isinstance(obj, ConfigDict):
            return obj
        if isinstance(obj, dict):
            return ConfigDict({k: ConfigDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(ConfigDict.wrap(v) for v in obj)
        return obj


class Config(ConfigDict):
    """`Config.fromfile(path)` executes a python config file, like mmcv.Config."""

    @staticmethod
    def fromfile(path):
        scope = {'__file__': os.path.abspath(path)}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), scope)
        cfg = Config(ConfigDict.wrap({k: v for k, v in scope.items()
                                      if not k.startswith('__') and not callable(v)
                                      and not isinstance(v, type(os))}))
        cfg['filename'] = path
        return cfg

    def merge_from_dict(self, options):
        """--cfg-options key.sub=value overrides (tools/train.py:72-81)."""
        for key, value in options.items():
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                if p not in node or not isinstance(node[p], dict):
                    node[p] = ConfigDict()          # mmcv creates the intermediate dicts (e.g. fp16.loss_scale=512.)
                node = node[p]
            node[parts[-1]] = value


class Registry:
    def __init__(self, name, parent=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return self.get(key) is not None

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        return self.parent.get(key) if self.parent is not None else None

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            names = [name] if isinstance(name, str) else (name or [cls.__name__])
            for n in names:
                if n in self._module_dict and not force:
                    raise KeyError(f'{n} is already registered in {self._name}')
                self._module_dict[n] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    cls = registry.get(obj_type) if isinstance(obj_type, str) else obj_type
    if cls is None:
        raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    return cls(**args)
