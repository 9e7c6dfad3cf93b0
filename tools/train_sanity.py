"""a few hundred steps on random frames + random messages: the losses must stay finite and the bit error must fall.
usage: train_sanity.py [size] [steps] [Identity|Jpeg] [bf16|f32]   (f32 = the exact parity path: the two dtypes should track each other)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd.hidden_models import Hidden
from video_watermarking_forgery_detection_amd import noise_layers as NL
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
dev = torch.device("cuda", 0)
torch.manual_seed(10)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
noise = NL.Identity() if (len(sys.argv) > 3 and sys.argv[3] == "Identity") else NL.Jpeg(50)
dt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.bfloat16
h = Hidden(HiDDenConfiguration(H=S, W=S), dev, noise, None, compute_dtype=dt)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for it in range(N):
    images = torch.rand(16, 3, S, S, device=dev)
    messages = torch.randint(0, 2, (16, 30), device=dev).float()
    losses, _ = h.train_on_batch([images, messages])
    if it % 50 == 0 or it == N - 1:
        print(it, {k.strip(): round(v, 4) for k, v in losses.items()})
    assert all(v == v and abs(v) < 1e4 for v in losses.values()), losses
