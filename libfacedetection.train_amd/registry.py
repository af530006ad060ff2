"""mmcv-compatible Registry / build_from_cfg / ConfigDict / Config.

The reference resolves every class on the hot path through mmcv registries named in the
python-dict configs (mmdet/models/builder.py:7-15, mmdet/core/bbox/builder.py:4-16,
mmdet/core/anchor/builder.py:6-12).  mmcv is un-vendored and not installable here, so the
same surface is provided natively; only the behaviour the shipped configs use is kept.
"""
import argparse
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

    @property
    def pretty_text(self):
        """The configuration as python source (what mmcv.Config.pretty_text is for: the text the tools log and
        store in checkpoint meta; mmcv runs its output through a code formatter, so only the CONTENT agrees):
        one `name = value` statement per top-level key, dicts as dict(...) calls, loadable by Config.fromfile."""
        def fmt(v, ind):
            pad = ' ' * (ind + 4)
            if isinstance(v, dict):
                if not v:
                    return 'dict()'
                if all(isinstance(k, str) and k.isidentifier() for k in v):
                    body = ''.join(f'{pad}{k}={fmt(x, ind + 4)},\n' for k, x in v.items())
                    return 'dict(\n' + body + ' ' * ind + ')'
                body = ''.join(f'{pad}{k!r}: {fmt(x, ind + 4)},\n' for k, x in v.items())
                return '{\n' + body + ' ' * ind + '}'
            if isinstance(v, (list, tuple)):
                o, c = ('[', ']') if isinstance(v, list) else ('(', ')')
                if not any(isinstance(x, (dict, list, tuple)) for x in v):
                    inner = ', '.join(fmt(x, ind) for x in v)
                    return o + inner + (',' if isinstance(v, tuple) and len(v) == 1 else '') + c
                body = ''.join(f'{pad}{fmt(x, ind + 4)},\n' for x in v)
                return o + '\n' + body + ' ' * ind + c
            if isinstance(v, range):
                return repr(list(v))
            return repr(v)
        return ''.join(f'{k} = {fmt(v, 0)}\n' for k, v in self.items() if k != 'filename')

    def dump(self, path=None):
        """mmcv.Config.dump for python configs: writes pretty_text (tools/train.py:171 stores the run's
        configuration next to its logs); returns the text when no path is given."""
        text = self.pretty_text
        if path is None:
            return text
        with open(path, 'w') as f:
            f.write(text)


class DictAction(argparse.Action):
    """argparse action of `--cfg-options` / `--options` (mmcv.DictAction, used by tools/train.py:66-82 and
    tools/test_widerface.py): KEY=VALUE pairs -> dict.  A value is an int, a float, true / false (any case), or
    a string; `a,b`, `[a,b]` give lists and `(a,b)` a tuple, nested to any depth; quotes around the value and
    blanks inside it are dropped."""

    @staticmethod
    def _scalar(text):
        for conv in (int, float):
            try:
                return conv(text)
            except ValueError:
                pass
        if text.lower() in ('true', 'false'):
            return text.lower() == 'true'
        return text

    @classmethod
    def parse_value(cls, text):
        text = text.strip('\'"').replace(' ', '')
        as_tuple = False
        if text.startswith('(') and text.endswith(')'):
            as_tuple, text = True, text[1:-1]
        elif text.startswith('[') and text.endswith(']'):
            text = text[1:-1]
        elif ',' not in text:
            return cls._scalar(text)
        items, depth, start = [], 0, 0
        for i, ch in enumerate(text):
            if ch in '([':
                depth += 1
            elif ch in ')]':
                depth -= 1
                if depth < 0:
                    raise ValueError(f'unbalanced brackets in {text!r}')
            elif ch == ',' and depth == 0:
                items.append(text[start:i])
                start = i + 1
        if depth != 0:
            raise ValueError(f'unbalanced brackets in {text!r}')
        if start < len(text):
            items.append(text[start:])
        out = [cls.parse_value(t) for t in items]
        return tuple(out) if as_tuple else out

    def __call__(self, parser, namespace, values, option_string=None):
        options = {}
        for kv in values:
            if '=' not in kv:
                raise argparse.ArgumentError(self, f'expected KEY=VALUE, got {kv!r}')
            key, val = kv.split('=', 1)
            options[key] = self.parse_value(val)
        setattr(namespace, self.dest, options)


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
