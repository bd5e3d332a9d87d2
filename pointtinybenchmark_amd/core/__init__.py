from .assigners import (AssignResult, DisCostV2, FocalLossCost, HungarianAssignerV2, PointAssigner,  # noqa: F401
                        PseudoSampler, SamplingResult)
from .point_generator import PointGenerator  # noqa: F401
from .post_processing import multiclass_nms  # noqa: F401
