"""TEST INFRASTRUCTURE ONLY -- loader that runs the *unmodified* reference files.

The reference (/root/reference, an MMDetection-2.24.1 fork) needs `mmcv-full`, which
is not installable here.  Its YuNet training path never calls a compiled mmcv op, so
the hot-path files execute unchanged under an arithmetic-free stub of `mmcv`
(Registry / build_from_cfg / identity decorators / BaseModule) plus `sys.modules`
package skeletons whose ``__path__`` points at the real reference directories (so
the heavy mmdet ``__init__`` zoo imports are skipped).  See SURVEY.md Appendix A.

This module exists to (1) pin `oracle/yunet_oracle.py` against the reference itself
and (2) generate the committed fixtures under `tests/golden/` (`oracle/make_golden.py`).
It works where /root/reference exists (the build container) or where oracle/make_ref.sh
has put the hot-path files under the git-ignored oracle/_ref/ (bench.py's cpu_baseline on
the GPU box); no test on the GPU box imports it.  It contains no reference source text:
every reference module is imported from where it lies.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

# /root/reference where it exists (the build container); otherwise oracle/_ref -- the git-ignored copy of the hot-path
# files made by oracle/make_ref.sh, which travels to the GPU box so that bench.py can time the reference there
_LOCAL_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')
REF_ROOT = os.environ.get('YUNET_REFERENCE_ROOT') or ('/root/reference' if os.path.isdir('/root/reference/mmdet') else _LOCAL_REF)


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'mmdet'))


# --------------------------------------------------------------------------- mmcv stub
class _Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def _register(self, cls, name=None, force=False):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, *args, **kwargs):
        default_args = kwargs.get('default_args')
        return _build_from_cfg(cfg, self, default_args)


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    cls = registry.get(obj_type) if isinstance(obj_type, str) else obj_type
    if cls is None:
        raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    return cls(**args)


class ConfigDict(dict):
    """Attribute-access dict (stand-in for mmcv.ConfigDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return ConfigDict({k: ConfigDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(ConfigDict.wrap(v) for v in obj)
        return obj


def _identity_decorator_factory(*a, **kw):
    if len(a) == 1 and callable(a[0]) and not kw:
        return a[0]

    def deco(f):
        return f
    return deco


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self._is_init = False

    def init_weights(self):
        self._is_init = True


def _raise(*a, **k):
    raise RuntimeError('compiled mmcv op is not available in the oracle stub')


def _install_mmcv_stub():
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_yunet_stub', False):
        return
    mmcv = types.ModuleType('mmcv')
    mmcv._yunet_stub = True
    mmcv.__version__ = '1.3.17'
    mmcv.jit = _identity_decorator_factory
    mmcv.ConfigDict = ConfigDict
    mmcv.is_str = lambda s: isinstance(s, str)

    utils = types.ModuleType('mmcv.utils')
    utils.Registry = _Registry
    utils.build_from_cfg = _build_from_cfg
    utils.print_log = lambda *a, **k: None
    utils.ConfigDict = ConfigDict

    cnn = types.ModuleType('mmcv.cnn')
    cnn.MODELS = _Registry('model')
    cnn_utils = types.ModuleType('mmcv.cnn.utils')
    weight_init = types.ModuleType('mmcv.cnn.utils.weight_init')
    weight_init.constant_init = lambda *a, **k: None
    cnn_utils.weight_init = weight_init
    cnn.utils = cnn_utils

    runner = types.ModuleType('mmcv.runner')
    runner.BaseModule = _BaseModule
    runner.force_fp32 = _identity_decorator_factory
    runner.auto_fp16 = _identity_decorator_factory
    runner.get_dist_info = lambda: (0, 1)
    runner.OptimizerHook = object

    ops = types.ModuleType('mmcv.ops')
    ops.batched_nms = _raise
    ops_nms = types.ModuleType('mmcv.ops.nms')
    ops_nms.batched_nms = _raise
    ops.nms = ops_nms

    mmcv.utils, mmcv.cnn, mmcv.runner, mmcv.ops = utils, cnn, runner, ops
    for name, mod in [('mmcv', mmcv), ('mmcv.utils', utils), ('mmcv.cnn', cnn),
                      ('mmcv.cnn.utils', cnn_utils),
                      ('mmcv.cnn.utils.weight_init', weight_init),
                      ('mmcv.runner', runner), ('mmcv.ops', ops),
                      ('mmcv.ops.nms', ops_nms)]:
        sys.modules[name] = mod


_PKGS = [
    'mmdet', 'mmdet.core', 'mmdet.core.bbox', 'mmdet.core.bbox.assigners',
    'mmdet.core.bbox.samplers', 'mmdet.core.bbox.iou_calculators',
    'mmdet.core.anchor', 'mmdet.core.utils', 'mmdet.core.mask', 'mmdet.models',
    'mmdet.models.utils', 'mmdet.models.losses', 'mmdet.models.dense_heads',
    'mmdet.models.backbones', 'mmdet.models.necks', 'mmdet.models.detectors',
    'mmdet.utils',
]

_LEAVES = [
    'mmdet.utils.util_mixins',
    'mmdet.core.bbox.builder',
    'mmdet.core.bbox.iou_calculators.builder',
    'mmdet.core.bbox.iou_calculators.iou2d_calculator',
    'mmdet.core.bbox.assigners.assign_result',
    'mmdet.core.bbox.assigners.base_assigner',
    'mmdet.core.bbox.assigners.sim_ota_assigner',
    'mmdet.core.bbox.samplers.sampling_result',
    'mmdet.core.bbox.samplers.base_sampler',
    'mmdet.core.bbox.samplers.pseudo_sampler',
    'mmdet.core.anchor.builder',
    'mmdet.core.anchor.point_generator',
    'mmdet.core.utils.misc',
    'mmdet.core.utils.dist_utils',
    'mmdet.models.builder',
    'mmdet.models.utils.yunet_layer',
    'mmdet.models.losses.utils',
    'mmdet.models.losses.cross_entropy_loss',
    'mmdet.models.losses.iou_loss',
    'mmdet.models.losses.smooth_l1_loss',
    'mmdet.models.backbones.yunet_backbone',
    'mmdet.models.necks.tfpn',
    'mmdet.models.dense_heads.base_dense_head',
    'mmdet.models.dense_heads.dense_test_mixins',
    'mmdet.models.dense_heads.yunet_head',
    'mmdet.models.detectors.base',
    'mmdet.models.detectors.single_stage',
    'mmdet.models.detectors.yunet',
]

_loaded = None


def load_reference():
    """Import the reference hot-path modules; returns a namespace of handles."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f'reference tree not found at {REF_ROOT}')
    _install_mmcv_stub()
    for name in _PKGS:
        if name in sys.modules:
            continue
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF_ROOT, *name.split('.'))]
        mod.__package__ = name
        sys.modules[name] = mod
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(sys.modules[parent], child, mod)

    # core/mask/structures.py pulls cv2 / pycocotools / mmcv.ops.roi_align; the
    # training path only needs the two class names for an isinstance check.
    ms = types.ModuleType('mmdet.core.mask.structures')
    ms.BitmapMasks = type('BitmapMasks', (), {})
    ms.PolygonMasks = type('PolygonMasks', (), {})
    sys.modules['mmdet.core.mask.structures'] = ms
    sys.modules['mmdet.core.mask'].structures = ms

    mu = sys.modules['mmdet.utils']
    mu.get_root_logger = lambda *a, **k: __import__('logging').getLogger('mmdet')
    ctxm = types.ModuleType('mmdet.utils.contextmanagers')
    ctxm.completed = None
    sys.modules['mmdet.utils.contextmanagers'] = ctxm

    core = sys.modules['mmdet.core']
    core_utils = sys.modules['mmdet.core.utils']
    # names yunet_head / base_dense_head import from the package level
    core_utils.filter_scores_and_topk = None
    core_utils.select_single_mlvl = None
    core.bbox2result = None
    core.bbox_mapping_back = None
    core.merge_aug_proposals = None

    mods = {}
    for leaf in _LEAVES:
        mods[leaf] = importlib.import_module(leaf)
        if leaf == 'mmdet.core.bbox.iou_calculators.iou2d_calculator':
            sys.modules['mmdet.core.bbox.iou_calculators'].bbox_overlaps = \
                mods[leaf].bbox_overlaps
            core.bbox_overlaps = mods[leaf].bbox_overlaps
        if leaf == 'mmdet.core.bbox.builder':
            core.build_assigner = mods[leaf].build_assigner
            core.build_sampler = mods[leaf].build_sampler
        if leaf == 'mmdet.core.utils.misc':
            core.multi_apply = mods[leaf].multi_apply
        if leaf == 'mmdet.core.utils.dist_utils':
            core.reduce_mean = mods[leaf].reduce_mean

    ns = types.SimpleNamespace()
    ns.mods = mods
    ns.builder = mods['mmdet.models.builder']
    ns.MODELS = ns.builder.MODELS
    ns.ConfigDict = ConfigDict
    ns.bbox_overlaps = mods['mmdet.core.bbox.iou_calculators.iou2d_calculator'].bbox_overlaps
    ns.SimOTAAssigner = mods['mmdet.core.bbox.assigners.sim_ota_assigner'].SimOTAAssigner
    ns.MlvlPointGenerator = mods['mmdet.core.anchor.point_generator'].MlvlPointGenerator
    ns.ConvDPUnit = mods['mmdet.models.utils.yunet_layer'].ConvDPUnit
    ns.Conv_head = mods['mmdet.models.utils.yunet_layer'].Conv_head
    ns.Conv4layerBlock = mods['mmdet.models.utils.yunet_layer'].Conv4layerBlock
    ns.losses = types.SimpleNamespace(
        ce=mods['mmdet.models.losses.cross_entropy_loss'],
        iou=mods['mmdet.models.losses.iou_loss'],
        sl1=mods['mmdet.models.losses.smooth_l1_loss'],
        utils=mods['mmdet.models.losses.utils'])
    _loaded = ns
    return ns


def load_config(name):
    """exec() a reference python config (configs/yunet_{n,s}.py) -> ConfigDict."""
    path = name if os.path.isabs(name) else os.path.join(REF_ROOT, 'configs', name)
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), scope)
    return ConfigDict.wrap({k: v for k, v in scope.items() if not k.startswith('__')})


def build_detector(cfg_name='yunet_n.py', loss_bbox=None, head=None):
    """Build the reference YuNet detector (train mode, CPU).  `head`: bbox_head settings to override (a value of None
    deletes the key, so that the reference class's own default applies)."""
    ns = load_reference()
    cfg = load_config(cfg_name)
    model_cfg = cfg.model
    if loss_bbox is not None:
        model_cfg.bbox_head.loss_bbox = ConfigDict.wrap(loss_bbox)
    for k, v in (head or {}).items():
        if v is None:
            model_cfg.bbox_head.pop(k, None)
        else:
            model_cfg.bbox_head[k] = v
    train_cfg = model_cfg.pop('train_cfg')
    test_cfg = model_cfg.pop('test_cfg')
    model_cfg['train_cfg'] = train_cfg
    model_cfg['test_cfg'] = test_cfg
    model = ns.builder.MODELS.build(model_cfg)
    model.train()
    return model, cfg


def load_checkpoint_state(name):
    ck = torch.load(os.path.join(REF_ROOT, 'weights', name), map_location='cpu',
                    weights_only=False)
    return ck['state_dict'] if 'state_dict' in ck else ck


if __name__ == '__main__':
    m, _ = build_detector('yunet_n.py')
    print('params', sum(p.numel() for p in m.parameters()))
    m.load_state_dict(load_checkpoint_state('yunet_n.pth'), strict=True)
    print('ckpt loaded strict')


# --------------------------------------------------------------------------- data pipeline
def load_pipeline_transforms(imresize=None, imflip=None):
    """Import the UNMODIFIED reference `mmdet/datasets/pipelines/transforms.py` (RandomSquareCrop,
    Resize, RandomFlip, Normalize) for pinning `oracle/pipeline_oracle.py`.

    Its module-level imports need cv2 (absent here) and mmcv image ops; both are stubbed:
    `mmcv.imresize` / `mmcv.imflip` are supplied by the caller (the geometry, RNG call order and
    box / keypoint arithmetic all run in the reference's own code; only the pixel interpolation of
    cv2.resize is outside what can be executed here).  `np.int`, removed from numpy >= 1.24 and
    used at transforms.py:1073, is restored as an alias of `int`.
    """
    import numpy as np
    load_reference()
    if not hasattr(np, 'int'):
        np.int = int
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')
    mmcv = sys.modules['mmcv']
    mmcv.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq)
    mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(x, t) for x in seq)
    if imresize is not None:
        mmcv.imresize = imresize
    if imflip is not None:
        mmcv.imflip = imflip
    mmcv.imnormalize = lambda img, mean, std, to_rgb=True: ((img - mean) / std).astype(np.float32)
    for name in ('mmdet.datasets', 'mmdet.datasets.pipelines', 'mmdet.core.evaluation'):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = [os.path.join(REF_ROOT, *name.split('.'))]
            mod.__package__ = name
            sys.modules[name] = mod
            parent, child = name.rsplit('.', 1)
            setattr(sys.modules[parent], child, mod)
    if 'mmdet.datasets.builder' not in sys.modules:
        b = types.ModuleType('mmdet.datasets.builder')
        b.PIPELINES = _Registry('pipeline')
        sys.modules['mmdet.datasets.builder'] = b
        sys.modules['mmdet.datasets'].builder = b
    core = sys.modules['mmdet.core']
    ms = sys.modules['mmdet.core.mask.structures']
    core.BitmapMasks, core.PolygonMasks = ms.BitmapMasks, ms.PolygonMasks
    core.find_inside_bboxes = None
    sys.modules['mmdet.utils'].log_img_scale = lambda *a, **k: False
    importlib.import_module('mmdet.core.evaluation.bbox_overlaps')
    return importlib.import_module('mmdet.datasets.pipelines.transforms')
