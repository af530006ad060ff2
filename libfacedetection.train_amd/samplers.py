"""Sample order of an epoch: the reference's group samplers (mmdet/datasets/samplers/group_sampler.py), which
`build_dataloader` puts in front of every training DataLoader (mmdet/datasets/builder.py:87-190:
DistributedGroupSampler(dataset, samples_per_gpu, world_size, rank, seed) when distributed, GroupSampler otherwise).

Both keep the `samples_per_gpu` images of one batch inside one aspect-ratio group (`dataset.flag`: 1 for
width / height > 1, CustomDataset._set_group_flag) and pad every group to whole batches by repeating samples.

  DistributedGroupSampler  torch.Generator seeded with seed + epoch: one randperm per non-empty group (the group is
                           then repeated cyclically up to a multiple of samples_per_gpu * num_replicas), one randperm
                           over the batch-sized blocks of the concatenation; rank r takes the r-th contiguous part.
                           Deterministic: `tests/test_samplers.py` pins the order to the unmodified reference class.
  GroupSampler             the same idea on numpy's GLOBAL generator (shuffle, choice, permutation): reproducible only
                           as far as nothing else draws from np.random in between; the same calls in the same order
                           are made here, so under np.random.seed(s) both give the same epoch.

Host-side index arithmetic only; the images these indices select are decoded on the host and augmented on the
device (datasets.RetinaFaceSource).
"""
import math

import numpy as np
import torch


def _group_members(flag):
    flag = np.asarray(flag).astype(np.int64)
    return [(v, np.flatnonzero(flag == v)) for v in range(int(flag.max()) + 1 if flag.size else 0)]


class GroupSampler:
    def __init__(self, dataset, samples_per_gpu=1):
        if not hasattr(dataset, 'flag'):
            raise AssertionError('GroupSampler needs dataset.flag (the aspect-ratio group of every sample)')
        self.dataset, self.samples_per_gpu = dataset, int(samples_per_gpu)
        self.flag = np.asarray(dataset.flag).astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        b = self.samples_per_gpu
        self.num_samples = int(sum(-(-int(s) // b) * b for s in self.group_sizes))

    def __iter__(self):
        b = self.samples_per_gpu
        parts = []
        for _, members in _group_members(self.flag):
            if members.size == 0:
                continue
            np.random.shuffle(members)
            short = -(-members.size // b) * b - members.size
            parts.append(np.concatenate([members, np.random.choice(members, short)]))
        flat = np.concatenate(parts)
        order = np.random.permutation(range(flat.size // b))
        return iter(flat.reshape(-1, b)[order].reshape(-1).astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler:
    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if on else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if on else 0) if rank is None else rank
        if not hasattr(dataset, 'flag'):
            raise AssertionError('DistributedGroupSampler needs dataset.flag')
        self.dataset, self.samples_per_gpu = dataset, int(samples_per_gpu)
        self.num_replicas, self.rank, self.epoch = int(num_replicas), int(rank), 0
        self.seed = seed if seed is not None else 0
        self.flag = np.asarray(dataset.flag)
        self.group_sizes = np.bincount(self.flag.astype(np.int64))
        per_rank_batches = sum(math.ceil(int(s) / self.samples_per_gpu / self.num_replicas) for s in self.group_sizes)
        self.num_samples = int(per_rank_batches * self.samples_per_gpu)
        self.total_size = self.num_samples * self.num_replicas

    def epoch_order(self):
        """The whole epoch, all ranks: rank r owns [r * num_samples, (r + 1) * num_samples)."""
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        b, chunk = self.samples_per_gpu, self.samples_per_gpu * self.num_replicas
        parts = []
        for _, members in _group_members(self.flag):
            if members.size == 0:
                continue
            shuffled = members[torch.randperm(int(members.size), generator=g).numpy()]
            parts.append(np.resize(shuffled, -(-members.size // chunk) * chunk))     # cyclic repetition
        flat = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        assert flat.size == self.total_size
        blocks = torch.randperm(flat.size // b, generator=g).numpy()
        return flat.reshape(-1, b)[blocks].reshape(-1).astype(np.int64)

    def __iter__(self):
        lo = self.num_samples * self.rank
        return iter(self.epoch_order()[lo:lo + self.num_samples].tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
