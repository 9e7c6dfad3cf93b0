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
    """Gradient all-reduce of flat f32 buckets: SUM over the ranks on RCCL's stream, the division by the world size folded into
    the optimiser kernel (`scale`, handed to wm_adam_step as grad_scale) instead of a pass of its own.

    start(bucket) queues the collective behind the work already on the current stream and returns at once; finish(handle)
    makes the current stream wait for it.  A bucket is any contiguous slice of a network's flat gradient buffer, so a network
    can go out in several reverse-order buckets while the rest of its backward still runs:
      * discriminator (0.30 MB): started after its second backward, travels under the decoder's forward;
      * decoder (0.97 MB): started after its backward, travels under the attack's and the encoder's backward;
      * encoder (0.68 MB): [after_concat, final] first, under the four body layers' backward, then [conv_layers];
      * UNet (31 MB): four buckets in reverse layer order, started from inside its backward.
    callable(bucket) = start + finish + the division applied in place (for callers that read the averaged gradients).
    `force` runs the collectives at world size 1 too (tests exercise the exact multi-rank code path on one GPU)."""

    def __init__(self, group=None, force=False, profile=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.scale = 1.0 / self.world
        self.profile = profile      # record, per bucket, its bytes and how long the compute stream stood waiting for it (report())
        self._waits = []

    def start(self, flat):
        if not self.active:
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat

    def finish(self, handle):
        if handle is None:
            return
        if self.profile and handle[1].is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            handle[0].wait()        # the current stream waits for the collective: e1 - e0 is the part of it that did NOT travel under compute
            e1.record()
            self._waits.append((handle[1].numel() * handle[1].element_size(), e0, e1))
        else:
            handle[0].wait()

    def report(self, steps=1):
        """(profile=True) -> {"buckets_per_step", "bytes_per_step", "bucket_bytes": the distinct sizes, "wait_ms_per_step": the time the
        compute stream stood in finish() -- the exposed, non-overlapped part of the all-reduces --, "wait_ms_max_bucket"}; clears the log"""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        w, self._waits = self._waits, []
        ms = [e0.elapsed_time(e1) for _, e0, e1 in w]
        steps = max(int(steps), 1)
        return {"world": self.world, "buckets_per_step": len(w) / steps, "bytes_per_step": sum(b for b, _, _ in w) / steps,
                "bucket_bytes": sorted({b for b, _, _ in w}), "wait_ms_per_step": sum(ms) / steps, "wait_ms_max_bucket": max(ms) if ms else 0.0}

    def finish_all(self, handles):
        for h in handles:
            self.finish(h)

    def average_(self, flat):
        """the division, in place (only where something other than the optimiser reads the gradients: clipping)"""
        if self.world > 1:
            flat.mul_(self.scale)
        return flat

    def __call__(self, flat):
        self.finish(self.start(flat))
        return self.average_(flat)


class CallableGradSync:
    """a plain callable(flat_grad_tensor) -- expected to leave the AVERAGED gradient in the tensor -- behind GradSync's interface: it runs
    synchronously where the bucket would be waited for; nothing is folded into the optimiser (scale 1)"""
    scale = 1.0
    world = 1
    active = True

    def __init__(self, fn):
        self.fn = fn

    def start(self, flat):
        return flat

    def finish(self, handle):
        if handle is not None:
            self.fn(handle)

    def finish_all(self, handles):
        for h in handles:
            self.finish(h)

    def average_(self, flat):
        return flat

    def __call__(self, flat):
        self.fn(flat)
        return flat


def as_grad_sync(obj):
    """None, a GradSync-like object (has start / finish), or a plain callable -> the interface Hidden / UNet drive"""
    if obj is None or hasattr(obj, "start"):
        return obj
    if callable(obj):
        return CallableGradSync(obj)
    raise TypeError("grad_sync must be None, a distributed.GradSync or a callable(flat_grad_tensor)")


def broadcast_parameters(modules, src=0, group=None):
    """DDP's wrap-time broadcast: parameters and buffers of rank `src` to every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        if hasattr(m, "flat_params"):
            dist.broadcast(m.flat_params, src, group=group)
            if hasattr(m, "invalidate_packs"):
                m.invalidate_packs()      # the parameters changed behind every version counter
        else:
            for p in m.parameters():
                dist.broadcast(p.data, src, group=group)
        for b in m.buffers():
            dist.broadcast(b, src, group=group)
