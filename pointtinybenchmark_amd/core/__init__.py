from .assigners import (AssignResult, DisCostV2, FocalLossCost, HungarianAssignerV2, PointAssigner,  # noqa: F401
                        PseudoSampler, SamplingResult)
from .point_generator import PointGenerator  # noqa: F401
