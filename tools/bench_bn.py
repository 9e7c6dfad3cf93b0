"""micro-benchmark of the BatchNorm-backward streaming passes at the bench shape"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops, _lib
B, H, W, C = 16, 256, 256, 64
dev = "cuda"
y = torch.randn(B, H, W, C, device=dev).bfloat16(); g = torch.randn(B, H, W, C, device=dev).bfloat16()
stats = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5]).contiguous()
part = torch.empty(4096, 2, C, device=dev)
L = _lib.lib()
def red():
    L.wm_bn_bwd_reduce(ctypes.c_void_p(g.data_ptr()), C, None, ctypes.c_void_p(y.data_ptr()), C, ctypes.c_void_p(stats[0].data_ptr()), ctypes.c_void_p(stats[1].data_ptr()),
                       ctypes.c_void_p(stats[2].data_ptr()), ctypes.c_void_p(stats[3].data_ptr()), ctypes.c_void_p(part.data_ptr()), B, ctypes.c_size_t(H * W), C, 1,
                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
t = timeit(red)
print(f"WM_BNR_EXP={os.environ.get('WM_BNR_EXP')}: reduce {t:.1f} us ({2*y.numel()*2/t/1e6:.2f} TB/s)")
