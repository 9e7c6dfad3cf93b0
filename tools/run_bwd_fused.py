"""run the fused backward kernel (csrc/bwd_ws.hip) N times on random operands of the benchmark shape: a target for rocprofv3
usage: python3 tools/run_bwd_fused.py [N] [variant]   (variant: wm_debug_bwd_variant of the debug library)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import _lib, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:
    _lib._lib = _lib.debug_lib()
    _lib.lib().wm_debug_bwd_variant(ctypes.c_int(int(sys.argv[2])))
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
g = torch.randn(B, H, W, C, device="cuda").to(dt); y = torch.randn(B, H, W, C, device="cuda").to(dt); xr = torch.randn(B, H, W, C, device="cuda").to(dt)
stats = torch.rand(4, C, device="cuda") + 0.5; coef = torch.rand(3, C, device="cuda") * 0.01; coef[0] += 1.0
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
dw = torch.zeros(C, C, 3, 3, device="cuda")
for _ in range(n):
    ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, sc, sh, dw, False, premasked=True)   # the form the step launches (11 of its 13 launches)
torch.cuda.synchronize()
print("done", n)
