"""MI355X-native CPR / P2PNet point-localization hot path.

Importing the package registers the drop-in classes under the reference's registry names
(BasicLocator, ResNet, FPN, CPRHead, P2PHead, MILLoss, HungarianAssignerV2, PointAssigner, PseudoSampler,
FocalLossCost, DisCostV2)."""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  RCCL opens queues of its
# own when a process group is created, and from then on two torch streams can land on ONE hardware queue: the backward's second
# stream (weight gradients beside the data-gradient chain, training.BackwardEngine) then runs strictly after the first.  Measured
# under a 1-rank torchrun (rocprofv3 kernel trace: every kernel of the step on a single queue): configs[4] training 119 img/s
# against 141 without a process group; with 8 hardware queues 138.  The variable is read when the HIP runtime initialises (first
# device call), so it is set here -- before this package touches the device -- unless the caller has chosen a value.
# It is a process-wide setting, so it is applied only when the caller has not chosen one AND HIP is not initialised yet (afterwards it
# would be read by nobody: training.BackwardEngine then warns, naming the import order); CPR_SET_HW_QUEUES=0 leaves the environment alone.
HW_QUEUES_SET_BY_PACKAGE = False


def _hip_initialised():
    try:
        import torch as _torch
        return bool(_torch.cuda.is_initialized())
    except Exception:
        return False


if 'GPU_MAX_HW_QUEUES' not in _os.environ and _os.environ.get('CPR_SET_HW_QUEUES', '1') != '0' and not _hip_initialised():
    _os.environ['GPU_MAX_HW_QUEUES'] = '8'
    HW_QUEUES_SET_BY_PACKAGE = True

from . import registry  # noqa: F401,E402
from .backbones import ResNet  # noqa: F401,E402
from .core import (AssignResult, BBoxL1Cost, ClassificationCost, ClassificationCostV2, DisCostV2, FocalLossCost,  # noqa: F401,E402
                   HungarianAssigner, HungarianAssignerV2, IoUCost, IoUCostV2, PointAssigner, PointGenerator, PseudoSampler, ZeroCost)
from .dense_heads import CPRHead, P2PHead  # noqa: F401,E402
from .detectors import BasicLocator  # noqa: F401,E402
from .losses import MILLoss  # noqa: F401,E402
from .necks import FPN  # noqa: F401,E402
from .registry import (BACKBONES, BBOX_ASSIGNERS, BBOX_SAMPLERS, DETECTORS, HEADS, LOSSES, MATCH_COST, NECKS,  # noqa
                       build_assigner, build_backbone, build_detector, build_head, build_loss, build_match_cost,
                       build_neck, build_sampler)

__version__ = '0.1.0'
