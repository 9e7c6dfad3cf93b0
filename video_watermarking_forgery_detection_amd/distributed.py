"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through torch.distributed
(backend "nccl" IS RCCL on ROCm).  Mirrors the reference's `init_dist` (train.py:20-33) and what its
DistributedDataParallel wrap does for this path (models/IRNrhi_model.py:163-168): broadcast of the
parameters from rank 0 at start, SUM all-reduce of the gradients divided by world size before each
optimiser step.  Because every network keeps its gradients in ONE flat f32 buffer, a network is one
collective (HiDDeN enc 0.68 MB + dec 0.97 MB + disc 0.30 MB: latency-bound, so fewer, larger
messages are what xGMI wants); BatchNorm statistics stay per-rank like the reference's plain
BatchNorm2d (no SyncBN).
"""
import os

import torch
import torch.distributed as dist


def init_dist(backend="nccl", **kwargs):
    """train.py:20-33: RANK from the launcher env, device = rank % num_gpus."""
    rank = int(os.environ["RANK"])
    if backend == "nccl":
        num_gpus = torch.cuda.device_count()
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank % max(1, num_gpus))))
    dist.init_process_group(backend=backend, **kwargs)
    return dist.get_world_size(), dist.get_rank()


def shard_batch_size(batch_size, world_size):
    """data/__init__.py:16-17: the global batch must split evenly."""
    if batch_size % world_size != 0:
        raise ValueError(f"batch_size {batch_size} is not divisible by world size {world_size}")
    return batch_size // world_size


def shard_indices(n, rank, world_size):
    """data/data_sampler.py:46-60: strided shard indices[rank::world]."""
    return list(range(n))[rank::world_size]


class GradSync:
    """callable(flat_grad): all-reduce SUM then / world, in place.  start() / finish() split the call so that a network's
    bucket travels over xGMI while the backward of the next network still runs (hidden.py: the decoder's bucket overlaps the
    attack + encoder backward): the collective is queued on RCCL's stream behind the work already on the current stream, and
    finish() makes the current stream wait for it."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def start(self, flat):
        if self.world == 1:
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat

    def finish(self, handle):
        if handle is None:
            return
        work, flat = handle
        work.wait()
        flat.mul_(1.0 / self.world)

    def __call__(self, flat):
        self.finish(self.start(flat))
        return flat


def broadcast_parameters(modules, src=0, group=None):
    """DDP's wrap-time broadcast: parameters and buffers of rank `src` to every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        if hasattr(m, "flat_params"):
            dist.broadcast(m.flat_params, src, group=group)
        else:
            for p in m.parameters():
                dist.broadcast(p.data, src, group=group)
        for b in m.buffers():
            dist.broadcast(b, src, group=group)
