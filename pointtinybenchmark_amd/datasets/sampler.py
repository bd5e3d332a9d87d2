"""DistributedGroupSampler (T/mmdet/datasets/samplers/group_sampler.py:51-148): the index stream every rank draws its
``samples_per_gpu`` images from -- images are grouped by aspect-ratio flag (custom.py:163-173), every group is shuffled
with a generator seeded by ``epoch + seed`` and padded (by repetition) to a multiple of samples_per_gpu * num_replicas, the
samples_per_gpu-sized chunks are shuffled again, and rank r takes the r-th contiguous slice.  Same constructor, ``__iter__``,
``__len__`` and ``set_epoch``; pinned against the reference class by tests/golden/data_side.json."""
import math

import numpy as np
import torch


def group_flags(data_infos):
    """custom.py:163-173: flag 1 for images wider than high, else 0."""
    return np.array([1 if info['width'] / info['height'] > 1 else 0 for info in data_infos], dtype=np.uint8)


class DistributedGroupSampler:
    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            ok = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if ok else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if ok else 0) if rank is None else rank
        self.dataset, self.samples_per_gpu, self.num_replicas, self.rank = dataset, samples_per_gpu, num_replicas, rank
        self.epoch, self.seed = 0, (seed if seed is not None else 0)
        self.flag = dataset.flag if hasattr(dataset, 'flag') else group_flags(dataset.data_infos)
        self.group_sizes = np.bincount(self.flag)
        per = samples_per_gpu * num_replicas
        self.num_samples = sum(int(math.ceil(n / per)) * samples_per_gpu for n in self.group_sizes)
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        per = self.samples_per_gpu * self.num_replicas
        indices = []
        for flag, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.where(self.flag == flag)[0]
            shuffled = members[torch.randperm(int(size), generator=g).numpy()].tolist()
            target = int(math.ceil(size / per)) * per
            padded = (shuffled * (target // size + 1))[:target]        # repeat the shuffled group until it is long enough
            indices.extend(padded)
        chunks = torch.randperm(len(indices) // self.samples_per_gpu, generator=g).tolist()
        indices = [indices[j] for c in chunks for j in range(c * self.samples_per_gpu, (c + 1) * self.samples_per_gpu)]
        lo = self.num_samples * self.rank
        return iter(indices[lo:lo + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
