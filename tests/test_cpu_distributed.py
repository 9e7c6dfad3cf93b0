"""CPU, world_size 2, gloo: the data-parallel host logic of distributed.py (what DDP does for the
reference, train.py:20-33, IRNrhi_model.py:163-168, data/__init__.py:16-17, data_sampler.py:46-60):
rank-0 parameter broadcast, flat-bucket gradient all-reduce (SUM / world), batch sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from video_watermarking_forgery_detection_amd.distributed import (GradSync, broadcast_parameters, init_dist,
                                                                     shard_batch_size, shard_indices)
    from video_watermarking_forgery_detection_amd.hidden_models import Discriminator
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    w, r = init_dist(backend="gloo")
    assert (w, r) == (world, rank)
    torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
    net = Discriminator(HiDDenConfiguration(H=16, W=16))
    net.flatten_parameters_()
    before = net.flat_params.clone()
    broadcast_parameters([net])
    gathered = [torch.empty_like(before) for _ in range(world)]
    dist.all_gather(gathered, net.flat_params)
    assert all(torch.equal(gathered[0], t) for t in gathered)        # every rank now holds rank 0's weights
    if rank != 0:
        assert not torch.equal(before, net.flat_params)
    # parameters are still views of the flat buffer after the in-place broadcast
    assert net.linear.weight.data_ptr() >= net.flat_params.data_ptr()
    # flat-bucket gradient all-reduce = mean over ranks
    net.flat_grads.copy_(torch.arange(net.flat_grads.numel(), dtype=torch.float32) * (rank + 1))
    GradSync()(net.flat_grads)
    expect = torch.arange(net.flat_grads.numel(), dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    assert torch.allclose(net.flat_grads, expect)
    assert torch.equal(net.linear.bias.grad, net.flat_grads[-1:])    # .grad views see the reduced values
    # BatchNorm buffers are broadcast too, then stay per-rank (no SyncBN in the reference)
    assert shard_batch_size(64, world) == 32
    try:
        shard_batch_size(63, world)
        raise AssertionError("indivisible batch must raise")
    except ValueError:
        pass
    idx = shard_indices(10, rank, world)
    assert idx == list(range(10))[rank::world]
    out.put((rank, float(net.flat_grads.sum())))
    dist.destroy_process_group()


def test_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(out.get(timeout=5) for _ in range(2))
    assert res[0][1] == res[1][1]
