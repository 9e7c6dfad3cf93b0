"""N benchmarked training steps (B=16, 256x256, Jpeg(50), bf16, dead discriminator gradients dropped) and nothing else: a target for rocprofv3.
usage: python3 tools/run_steps.py [N] [graph]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_watermarking_forgery_detection_amd.hidden_models import Hidden            # noqa: E402
from video_watermarking_forgery_detection_amd import noise_layers as NL              # noqa: E402
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
S, B = 256, 16
torch.manual_seed(10)
h = Hidden(HiDDenConfiguration(H=S, W=S), dev, NL.Jpeg(50), None, compute_dtype=torch.bfloat16, keep_dead_discriminator_grads=False)
if "two" in sys.argv:
    h.two_streams = True
if "graph" in sys.argv:
    h.enable_graph()
images = torch.rand(B, 3, S, S, device=dev)
messages = torch.randint(0, 2, (B, 30), device=dev).float()
for _ in range(n):
    losses, _ = h.train_on_batch([images, messages])
torch.cuda.synchronize()
print({k.strip(): v for k, v in losses.items()})
