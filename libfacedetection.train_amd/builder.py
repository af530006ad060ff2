"""Registries and builders with the reference's names (mmdet/models/builder.py:7-59,
mmdet/core/bbox/builder.py:4-21, mmdet/core/anchor/builder.py:6-19)."""
from .registry import Registry, build_from_cfg

MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = DETECTORS = MODELS
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
PRIOR_GENERATORS = Registry('Generator for anchors and points')
ANCHOR_GENERATORS = PRIOR_GENERATORS
PIPELINES = Registry('pipeline')          # mmdet/datasets/builder.py:23
DATASETS = Registry('dataset')            # mmdet/datasets/builder.py:22


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_prior_generator(cfg, default_args=None):
    return build_from_cfg(cfg, PRIOR_GENERATORS, default_args)


def build_dataset(cfg, default_args=None):
    """mmdet/datasets/builder.py:59 for the dataset types this package registers."""
    return build_from_cfg(cfg, DATASETS, default_args)
