"""Importable alias of the package directory ``libfacedetection.train_amd/``.

The directory name contains a dot (it mirrors the upstream project name), so it
cannot be imported directly.  This module turns itself into that package: its
``__path__`` points at the real directory, so ``import yunet_amd.engine`` etc.
resolve to ``libfacedetection.train_amd/engine.py``.
"""
import os as _os

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                     'libfacedetection.train_amd')
__path__ = [_dir]
__package__ = 'yunet_amd'
__file__ = _os.path.join(_dir, '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
del _f
