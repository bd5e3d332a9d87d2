"""Data side feeding the hot path: annotation parsing of CocoFmtDataset and the device tail of the
image pipeline (flip, normalise, pad, batch, channels-last)."""
from .cocofmt import CocoFmtDataset  # noqa: F401
from .pipeline import GpuImagePipeline  # noqa: F401
from .loader import BatchLoader  # noqa: F401
from .sampler import DistributedGroupSampler  # noqa: F401
from .tiles import generate_corner_dataset, image_tiles  # noqa: F401
