"""train.py -- launcher with the reference's command line (train.py:266-335):

    python -m torch.distributed.run --nproc_per_node=N -m video_watermarking_forgery_detection_amd.train \
        -opt options/train/train_hidden_c2.yml --launcher pytorch

`-opt <yml>`, `--launcher {none,pytorch}`, `--local_rank`, `-val {0,1}`.  The DAVIS loader of the
reference (data/) is outside the hot path: batches here are synthetic tensors of the loader's shape
contract ([B,3,H,W] frames or [B,3,T,H,W] clips + [B,1,T,H,W] masks), sharded as
data/__init__.py:16-17 does (per-rank batch = batch_size // world_size)."""
import argparse
import logging
import os
import random

import numpy as np
import torch

from .distributed import init_dist, shard_batch_size
from .models.IRNrhi_model import IRNrhiModel
from .options import options as option


def synthetic_batches(opt, per_rank_batch, rank, n):
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
    latest_values = None
    for train_data in synthetic_batches(opt, per_rank, rank, total_iters):
        current_step += 1
        if current_step > total_iters:
            break
        model.feed_data(train_data)
        if args.val == 0.0:
            logs, debug_logs = model.optimize_parameters(current_step, latest_values)
        else:
            logs, debug_logs = model.evaluate()
        if rank <= 0 and logs and current_step % 10 == 0:
            log.info("step %d  frames %d  %s", current_step, len(model.real_H),
                     "  ".join(f"{k}={v:.4f}" if isinstance(v, float) else f"{k}={v}" for k, v in logs))


if __name__ == '__main__':
    main()
