"""Batches for the hot path: sampler -> ``samples_per_gpu`` indices -> CocoFmtDataset.load_sample (decode, corner crop,
LoadAnnotations fields) -> GpuImagePipeline (flip / normalise / pad / collate on the device).  What the reference assembles
from build_dataloader + DistributedGroupSampler + collate (T/mmdet/datasets/builder.py:88-146, samplers/group_sampler.py,
mmcv.parallel.collate): one process per GPU, rank r draws the r-th slice of the shared shuffled index stream."""
import numpy as np

from .sampler import DistributedGroupSampler


class BatchLoader:
    def __init__(self, dataset, pipeline, samples_per_gpu=2, num_replicas=None, rank=None, seed=0, flip_seed=None):
        self.dataset, self.pipeline, self.samples_per_gpu = dataset, pipeline, samples_per_gpu
        self.sampler = DistributedGroupSampler(dataset, samples_per_gpu, num_replicas, rank, seed)
        # RandomFlip draws: the reference seeds every loader worker with num_workers * rank + worker_id + seed
        # (T/mmdet/datasets/builder.py worker_init_fn), i.e. the ranks draw DIFFERENT flip sequences; one in-process
        # "worker" per rank here -> seed + rank.  An explicit flip_seed is taken as given (tests).
        self.rng = np.random.RandomState(seed + self.sampler.rank if flip_seed is None else flip_seed)

    def set_epoch(self, epoch):
        self.sampler.set_epoch(epoch)

    def __len__(self):
        return len(self.sampler) // self.samples_per_gpu

    def __iter__(self):
        idx = list(iter(self.sampler))
        for i in range(0, len(idx), self.samples_per_gpu):
            samples = [self.dataset.load_sample(j) for j in idx[i:i + self.samples_per_gpu]]
            yield self.pipeline(samples, self.rng)
