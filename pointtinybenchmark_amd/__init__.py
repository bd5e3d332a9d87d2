"""MI355X-native CPR / P2PNet point-localization hot path.

Importing the package registers the drop-in classes under the reference's registry names
(BasicLocator, ResNet, FPN, CPRHead, P2PHead, MILLoss, HungarianAssignerV2, PointAssigner, PseudoSampler,
FocalLossCost, DisCostV2)."""
from . import registry  # noqa: F401
from .backbones import ResNet  # noqa: F401
from .core import (AssignResult, DisCostV2, FocalLossCost, HungarianAssignerV2, PointAssigner,  # noqa: F401
                   PointGenerator, PseudoSampler)
from .dense_heads import CPRHead, P2PHead  # noqa: F401
from .detectors import BasicLocator  # noqa: F401
from .losses import MILLoss  # noqa: F401
from .necks import FPN  # noqa: F401
from .registry import (BACKBONES, BBOX_ASSIGNERS, BBOX_SAMPLERS, DETECTORS, HEADS, LOSSES, MATCH_COST, NECKS,  # noqa
                       build_assigner, build_backbone, build_detector, build_head, build_loss, build_match_cost,
                       build_neck, build_sampler)

__version__ = '0.1.0'
