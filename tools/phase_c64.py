"""diagnostic: where the persistent 64-channel conv kernel spends its cycles (s_memtime stamps per phase)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops, _lib
B, H, W, C = 16, 256, 256, 64
x = torch.randn(B, H, W, C, device="cuda").bfloat16()
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wp = ops.pack_w3x3(w, C, C, torch.bfloat16)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
y = torch.empty_like(x)
names = ["barrier A (wait prev MFMA readers)", "halo write (transform + ds_write)", "barrier B", "prefetch + flush issue", "MFMA loop", "pack"]
for xf in (True, False):
    stamps = torch.zeros(256 * 4 * 6, dtype=torch.int64, device="cuda")
    L = _lib.lib()
    for _ in range(3):
        n = L.wm_debug_conv3x3_c64_phases(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                          ctypes.c_void_p(sc.data_ptr() if xf else 0), ctypes.c_void_p(sh.data_ptr() if xf else 0),
                                          ctypes.c_void_p(y.data_ptr()), B, H, W, ctypes.c_void_p(stamps.data_ptr()),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    s = stamps.view(256 * 4, 6).double().cpu()
    tiles = 16
    print(f"xform={xf}: cycles per tile per wave (median over {s.shape[0]} waves; s_memtime ticks = shader cycles at 100MHz*? see guide)")
    tot = 0
    for k, nm in enumerate(names):
        v = s[:, k].median().item() / tiles
        tot += v
        print(f"   {nm:40s} {v:9.0f}")
    print(f"   {'total':40s} {tot:9.0f}")
