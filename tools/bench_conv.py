"""micro-benchmark of the conv / wgrad kernels at the bench shape (B=16, 256x256, 64->64, bf16)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops

B, H, W, C = 16, 256, 256, 64
dev = "cuda"
torch.manual_seed(0)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
wp = ops.pack_w3x3(w, C, C, torch.bfloat16)
sc = torch.rand(C, device=dev) + 0.5
sh = torch.randn(C, device=dev) * 0.3
bias = torch.randn(C, device=dev) * 0.1
xs = {
    "normal": torch.randn(B, H, W, C, device=dev).bfloat16(),
    "relu": torch.relu(torch.randn(B, H, W, C, device=dev)).bfloat16(),
    "tiny": (torch.randn(B, H, W, C, device=dev) * 1e-6).bfloat16(),
    "zeros": torch.zeros(B, H, W, C, device=dev).bfloat16(),
}

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

flops = 2.0 * B * H * W * C * 9 * C
import ctypes
from video_watermarking_forgery_detection_amd import _lib
_lib._lib = _lib.debug_lib()   # the wm_debug_* switches exist only in the -DWM_DEBUG build (lib/libwm_hip_dbg.so)
for variant in [int(v) for v in os.environ.get('VARIANTS', '0').split(',')]:
  _lib.lib().wm_debug_ws_variant(ctypes.c_int(variant))
  for name, x in xs.items():
    t1 = timeit(lambda: ops.conv3x3_fwd(x, wp, bias, sc, sh, True))
    t2 = timeit(lambda: ops.conv3x3_fwd(x, wp, None, None, None, False))
    t3 = timeit(lambda: ops.conv3x3_fwd(x, wp, None, sc, sh, False))
    t4 = timeit(lambda: ops.conv3x3_fwd(x, wp, bias, None, None, True))
    print(f"variant {variant} conv c64 input={name:7s} xform+stats: {t1:7.1f} us  plain: {t2:7.1f} us  xform only: {t3:7.1f}  stats only: {t4:7.1f}")
_lib.lib().wm_debug_ws_variant(ctypes.c_int(0))
dw = torch.zeros(C, C, 3, 3, device=dev)
for name in ("normal", "relu"):
    x = xs[name]; dy = xs["normal"]
    t1 = timeit(lambda: ops.conv3x3_wgrad(x, C, sc, sh, dy, dw, False))
    t2 = timeit(lambda: ops.conv3x3_wgrad(x, C, None, None, dy, dw, False))
    print(f"wgrad x={name:7s} xform: {t1:7.1f} us ({flops/t1/1e6:6.0f} TF)   plain: {t2:7.1f} us ({flops/t2/1e6:6.0f} TF)")
