"""das_amd — MI355X-native (gfx950) implementation of the DAS data-parallel hot path.

Host side mirrors the reference's registry/config/method-name protocol (mmdet3d fork);
device side is libdas_hip.so (das_amd/csrc, C ABI in include/das_hip.h).
"""
from .config import Config, ConfigDict  # noqa: F401
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, Registry, build_backbone,  # noqa: F401
                       build_detector, build_from_cfg, build_head, build_loss, build_model, build_neck)
from . import losses  # noqa: F401  (registers the loss classes)
from .backbones import MSPN2  # noqa: F401
from .necks import FPN  # noqa: F401
from .pose_heads import DASHead  # noqa: F401
from .detectors import DAS  # noqa: F401

__version__ = '0.1.0'
