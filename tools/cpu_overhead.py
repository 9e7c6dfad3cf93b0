"""how much of the step is host launch time?  (enqueue time of N steps vs their GPU completion)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd.hidden_models import Hidden
from video_watermarking_forgery_detection_amd import noise_layers as NL
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
dev = torch.device("cuda", 0)
torch.manual_seed(10)
h = Hidden(HiDDenConfiguration(H=256, W=256), dev, NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
images = torch.rand(16, 3, 256, 256, device=dev)
messages = torch.randint(0, 2, (16, 30), device=dev).float()
for _ in range(5):
    h.train_on_batch([images, messages])
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    h.train_on_batch([images, messages])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step (train_on_batch ends with one host sync for the losses)")
# split: where the host time goes (cProfile of 5 steps)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    h.train_on_batch([images, messages])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
