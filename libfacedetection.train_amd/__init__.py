"""libfacedetection.train_amd -- the YuNet training hot path of
ShiqiYu/libfacedetection.train rebuilt for AMD MI355X (gfx950): hand-written HIP kernels
behind the reference's registry names.  Importing the package registers
YuNet / YuNetBackbone / TFPN / YuNet_Head / SimOTAAssigner / MlvlPointGenerator /
CrossEntropyLoss / EIoULoss / DIoULoss / SmoothL1Loss.
"""
from . import registry  # noqa: F401
from .builder import (BACKBONES, BBOX_ASSIGNERS, DATASETS, DETECTORS, HEADS, LOSSES, MODELS, NECKS,  # noqa
                      PRIOR_GENERATORS, build_assigner, build_backbone, build_dataset, build_detector,
                      build_head, build_loss, build_neck, build_prior_generator)
from .registry import Config, ConfigDict, Registry, build_from_cfg  # noqa: F401
from . import losses, point_generator, sim_ota_assigner  # noqa: F401,E402
from . import yunet_backbone, tfpn, yunet_head, yunet  # noqa: F401,E402
from . import pipelines  # noqa: F401,E402
from . import datasets, evaluation  # noqa: F401,E402
from .yunet import YuNet  # noqa: F401,E402

__version__ = '0.1.0'
