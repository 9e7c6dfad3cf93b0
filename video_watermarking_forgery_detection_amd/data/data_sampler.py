"""DistIterSampler: the iteration-oriented shard sampler train.py builds (reference: data/data_sampler.py:12-66; behaviour kept, text not).

Contract (what create_dataloader / train.py rely on):
  * the dataset is walked `ratio` times per "epoch": total = ceil(len(dataset) * ratio / world) * world draws, so the loader is restarted
    `ratio` times less often;
  * one permutation of range(total) is drawn from a fresh, UNSEEDED-by-epoch torch.Generator (the reference leaves upstream's
    `manual_seed(epoch)` out: every process draws the same default-seeded permutation, which is what keeps the ranks' shards disjoint),
    folded onto the dataset by `% len(dataset)`;
  * rank r keeps every world-th draw starting at r; len(sampler) = total / world; set_epoch only records the epoch.
"""
import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


def _world_and_rank(num_replicas, rank):
    """explicit values win; otherwise the initialised process group is asked (as the reference does: an error without one)"""
    if num_replicas is not None and rank is not None:
        return int(num_replicas), int(rank)
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("Requires distributed package to be available")
    return (dist.get_world_size() if num_replicas is None else int(num_replicas),
            dist.get_rank() if rank is None else int(rank))


class DistIterSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, ratio=100):
        self.dataset = dataset
        self.num_replicas, self.rank = _world_and_rank(num_replicas, rank)
        self.epoch = 0
        per_rank = -(-len(dataset) * ratio // self.num_replicas)        # ceil without floats
        self.num_samples = int(per_rank)
        self.total_size = self.num_samples * self.num_replicas

    def shard(self):
        """this rank's indices into the dataset for one pass, as an int64 tensor"""
        draws = torch.randperm(self.total_size, generator=torch.Generator())
        mine = draws[self.rank::self.num_replicas] % len(self.dataset)
        if mine.numel() != self.num_samples:
            raise AssertionError(f"shard of rank {self.rank} holds {mine.numel()} indices, expected {self.num_samples}")
        return mine

    def __iter__(self):
        return iter(self.shard().tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
