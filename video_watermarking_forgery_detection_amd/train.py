"""train.py -- launcher with the reference's command line (train.py:266-335):

    python -m torch.distributed.run --nproc_per_node=N -m video_watermarking_forgery_detection_amd.train \
        -opt options/train/train_hidden_c2.yml --launcher pytorch

`-opt <yml>`, `--launcher {none,pytorch}`, `--local_rank`, `-val {0,1}`.  With `datasets.train.dataroot` set, batches come from
the DAVIS clip loader (data/: `DVDataset` + `DistIterSampler` + `create_dataloader`, per-rank batch = batch_size // world_size as
data/__init__.py:16-17); without it they are synthetic tensors of the same shape contract ([B,3,H,W] frames or [B,3,T,H,W] clips +
[B,1,T,H,W] masks).  Rank 0 feeds the step's `logs` to the Progbar (train.py:93-96,109)."""
import argparse
import logging
import os
import random

import numpy as np
import torch

from .data import DistIterSampler, DVDataset, create_dataloader
from .distributed import init_dist, shard_batch_size
from .models.IRNrhi_model import DeferredLogs, IRNrhiModel
from .options import options as option
from .utils import Progbar


def davis_batches(opt, rank, world, n):
    """train.py:50-60,91-99 of the reference: dataset -> sampler -> loader, epochs until n iterations"""
    dopt = opt['datasets']['train']
    ds = DVDataset(root_path=dopt['dataroot'], image_size=dopt['GT_size'], clip_length=dopt['clip_length'])
    sampler = DistIterSampler(ds, world, rank, ratio=dopt['dist_ratio'] or 100) if opt['dist'] else None
    opt['phase'] = 'train'
    dopt['n_workers'] = dopt['n_workers'] or 0
    loader = create_dataloader(ds, dopt, opt, sampler)
    done = 0
    while done < n:
        for batch in loader:
            yield batch
            done += 1
            if done >= n:
                return


def synthetic_batches(opt, per_rank_batch, rank, n):
    """frames / clips of the loader's shape contract, drawn on ONE host thread: a multi-threaded torch.rand burst inside the training
    loop exhausts a cgroup CPU quota (a 16-CPU share of a 256-core host) and gets the whole process throttled for tens of ms"""
    torch.set_num_threads(1)
    size = opt['datasets']['train']['GT_size']
    T = opt['datasets']['train']['clip_length']
    g = torch.Generator().manual_seed((opt['train']['manual_seed'] or 10) + max(rank, 0))
    for _ in range(n):
        if T:
            imgs = torch.rand(per_rank_batch, 3, T, size, size, generator=g)
            mask = torch.zeros(per_rank_batch, 1, T, size, size)
            for b in range(per_rank_batch):
                hh, ww = int(size * 0.6 * torch.rand(1, generator=g)), int(size * 0.6 * torch.rand(1, generator=g))
                y0, x0 = int((size - hh) * torch.rand(1, generator=g)), int((size - ww) * torch.rand(1, generator=g))
                mask[b, :, :, y0:y0 + hh, x0:x0 + ww] = 1
            yield imgs, mask
        else:
            yield (torch.rand(per_rank_batch, 3, size, size, generator=g),)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True, help='Path to option YMAL file.')
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--local_rank', type=int, default=0)
    ap.add_argument('-val', type=float, default=0.0)
    args = ap.parse_args()
    opt = option.parse(args.opt, is_train=True)
    if args.launcher == 'none':
        opt['dist'] = False
        rank, world = -1, 1
    else:
        opt['dist'] = True
        world, rank = init_dist()
    seed = opt['train']['manual_seed'] or random.randint(1, 10000)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
    logging.basicConfig(level=logging.INFO if rank <= 0 else logging.WARNING, format="%(asctime)s %(message)s")
    log = logging.getLogger("base")
    model = IRNrhiModel(opt)
    per_rank = shard_batch_size(opt['datasets']['train']['batch_size'], world)
    current_step = opt['train']['current_step'] or 0
    total_iters = int(opt['train']['niter'])
    latest_values, pending = None, None
    progbar = Progbar(total_iters * per_rank, stateful_metrics=['lr', 'Kind', 'LocKind', 'PF']) if rank <= 0 and opt['train']['progbar'] else None
    batches = (davis_batches(opt, max(rank, 0), world, total_iters) if opt['datasets']['train']['dataroot']
               else synthetic_batches(opt, per_rank, rank, total_iters))
    for train_data in batches:
        current_step += 1
        if current_step > total_iters:
            break
        model.feed_data(train_data)
        if args.val == 0.0:
            logs, debug_logs = model.optimize_parameters(current_step, latest_values)
        else:
            logs, debug_logs = model.evaluate()
        if progbar is not None:
            if isinstance(logs, DeferredLogs):   # train.deferred_logs: the bar is fed one step late, after the next step has been enqueued
                if pending is not None:
                    progbar.add(pending[0], values=pending[1])
                pending = (len(model.real_H), logs)
            else:
                progbar.add(len(model.real_H), values=logs)      # train.py:109
        elif rank <= 0 and current_step % 10 == 0 and logs:
            log.info("step %d  frames %d  %s", current_step, len(model.real_H),
                     "  ".join(f"{k}={v:.4f}" if isinstance(v, float) else f"{k}={v}" for k, v in logs))
    if progbar is not None and pending is not None:
        progbar.add(pending[0], values=pending[1])


if __name__ == '__main__':
    main()
