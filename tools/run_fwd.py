"""run the forward 64 -> 64 conv (csrc/conv3x3_ws.hip, fused BN+ReLU input transform, BatchNorm statistics) N times on random operands of the
benchmark shape: a target for rocprofv3.    usage: python3 tools/run_fwd.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device="cuda").to(dt)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
bias = torch.randn(C, device="cuda") * 0.1
wp = ops.pack_w3x3(w, C, C, dt)
for _ in range(n):
    ops.conv3x3_fwd(x, wp, bias, sc, sh, want_stats=True)
torch.cuda.synchronize()
print("done", n)
