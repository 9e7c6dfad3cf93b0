"""ON THE GPU BOX: same-process A/B of engine.lazy_bn_finalize (the forward convolutions finalise their input's BatchNorm in their own
prologue) against the finalisation as a launch of its own, interleaved rounds, the benchmarked step (B=16, 256x256, Jpeg50, bf16).
usage: python tools/ab_lazy_bn.py [rounds] [two]   ("two": two chains on two streams)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_watermarking_forgery_detection_amd import engine                                # noqa: E402
from video_watermarking_forgery_detection_amd.hidden_models import Hidden                  # noqa: E402
from video_watermarking_forgery_detection_amd import noise_layers as NL                    # noqa: E402
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration           # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
dev = torch.device("cuda")
S, B = 256, 16
torch.manual_seed(10)
h = Hidden(HiDDenConfiguration(H=S, W=S), dev, NL.Jpeg(50), None, compute_dtype=torch.bfloat16, keep_dead_discriminator_grads=False)
h.two_streams = "two" in sys.argv
images = torch.rand(B, 3, S, S, device=dev)
messages = torch.randint(0, 2, (B, 30), device=dev).float()


def run(lazy, n=30):
    engine.lazy_bn_finalize(lazy)
    for _ in range(5):
        h.train_on_batch([images, messages])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        h.train_on_batch([images, messages])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


res = {True: [], False: []}
for r in range(rounds):
    for lazy in (False, True):
        res[lazy].append(run(lazy))
for lazy in (False, True):
    v = sorted(res[lazy])
    print(f"lazy_bn_finalize={lazy}: median {v[len(v) // 2]:.4f} ms/step  (all: {', '.join(f'{x:.4f}' for x in res[lazy])})")
