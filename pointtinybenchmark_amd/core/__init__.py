from .assigners import (AssignResult, BBoxL1Cost, ClassificationCost, ClassificationCostV2, DisCostV2, FocalLossCost,  # noqa: F401
                        HungarianAssigner, HungarianAssignerV2, IoUCost, IoUCostV2, PointAssigner, PseudoSampler, SamplingResult,
                        ZeroCost)
from .point_generator import PointGenerator  # noqa: F401
from .post_processing import multiclass_nms  # noqa: F401
