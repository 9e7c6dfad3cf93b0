"""per-phase cycle sums of the one-pass backward kernel (csrc/bwd_ws.hip, debug variant 4096 + 512: s_memtime stamps at the phase
boundaries of wave 0's tiles, summed per workgroup and written where dx would start).
usage: python3 tools/phase_bwd.py [extra variant bits; default 256 = the premasked form the step launches 10 times of 13]
(only the combinations instantiated in bwd_ws.hip's debug switch exist: 0, 8, 32, 64, 256; + 1048576 = the general addressing instead of
the whole-tile buffer-addressed form, which exists for 0, 8, 256)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import _lib, ops
_lib._lib = _lib.debug_lib()
variant = 4608 | (int(sys.argv[1]) if len(sys.argv) > 1 else 256)
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
g = torch.randn(B, H, W, C, device="cuda").to(dt); y = torch.randn(B, H, W, C, device="cuda").to(dt); xr = torch.randn(B, H, W, C, device="cuda").to(dt)
stats = torch.rand(4, C, device="cuda") + 0.5; coef = torch.rand(3, C, device="cuda") * 0.01; coef[0] += 1.0
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
dw = torch.zeros(C, C, 3, 3, device="cuda")
for _ in range(3):
    ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, sc, sh, dw, False)
_lib.lib().wm_debug_bwd_variant(ctypes.c_int(variant))
dx, part, _ = ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, sc, sh, dw, False)
torch.cuda.synchronize()
nwg = part.shape[0]
t = dx.view(torch.int64).reshape(-1)[: nwg * 8].reshape(nwg, 8).double().cpu()
ntiles = B * (H // 8) * (W // 16) / nwg
names = ["tile head (xr loads, addressing)", "weight-gradient loop", "vmcnt(0) wait", "input-gradient loop + staging", "epilogue", "barrier"]
m = t.mean(0)
print(f"variant {variant}: {nwg} workgroups x {ntiles:.0f} tiles; shader clock {m[6] / m[7] * 100:.0f} MHz (s_memtime / s_memrealtime x 100 MHz); run {m[7] / 100:.1f} us")
for i, n in enumerate(names):
    print(f"  {n:36s} {m[i] / ntiles:8.0f} cycles/tile  {100 * m[i] / m[:6].sum():5.1f} %")
print(f"  sum {m[:6].sum() / ntiles:.0f} cycles/tile; whole run {m[6] / ntiles:.0f} cycles/tile")
