"""Does walking a tensor backwards after the previous kernel walked it forwards hit the Infinity Cache?
chain of ws convs x0 -> x1 -> x2 ... at the bench shape; all-forward vs alternating direction."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops, _lib
_lib._lib = _lib.debug_lib()   # the wm_debug_* switches exist only in the -DWM_DEBUG build (lib/libwm_hip_dbg.so)

lib = _lib.lib()
B, H, W, C = 16, 256, 256, 64
dev = "cuda"
torch.manual_seed(0)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
wp = ops.pack_w3x3(w, C, C, torch.bfloat16)
sc = torch.rand(C, device=dev) + 0.5
sh = torch.randn(C, device=dev) * 0.3
bias = torch.randn(C, device=dev) * 0.1
x0 = torch.randn(B, H, W, C, device=dev).bfloat16()
NL = 6

def chain(alt):
    x = x0
    for l in range(NL):
        lib.wm_debug_ws_direction(int(alt and (l & 1)))
        x = ops.conv3x3_fwd(x, wp, bias, sc, sh, True)[0]
    lib.wm_debug_ws_direction(0)
    return x

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for rep in range(3):
    t0 = timeit(lambda: chain(False)); t1 = timeit(lambda: chain(True))
    print(f"chain of {NL} convs: forward {t0/NL:6.1f} us/conv   alternating {t1/NL:6.1f} us/conv")
ya = chain(False); yb = chain(True)
print("same result:", torch.equal(ya, yb))
