"""per-role cycle sums of the role-split one-pass backward kernel (csrc/bwd_ws8.hip, debug variant 1 << 21: s_memtime stamps of every wave,
written over the workgroup's partial rows).  W waves: MFMA loop | staging of the next a tile | wait at the tile's barrier;
D waves: MFMA loop with the dy staging between its MFMAs | wait at the barrier | epilogue.
usage: python3 tools/phase_bwd8.py [gvec]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import _lib, ops
_lib._lib = _lib.debug_lib()
gv = len(sys.argv) > 1 and sys.argv[1] == "gvec"
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
g = torch.randn(B, H, W, C, device="cuda").to(dt); y = torch.randn(B, H, W, C, device="cuda").to(dt); xr = torch.randn(B, H, W, C, device="cuda").to(dt)
stats = torch.rand(4, C, device="cuda") + 0.5; coef = torch.rand(3, C, device="cuda") * 0.01; coef[0] += 1.0
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
dw = torch.zeros(C, C, 3, 3, device="cuda")
gvec = torch.randn(B, C, device="cuda") / (H * W) if gv else None


def run():
    if gv:
        return ops.conv3x3_bwd_fused(None, y, stats, coef, wpt, xr, sc, sh, dw, False, gvec=gvec)
    return ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, sc, sh, dw, False, premasked=True)


for _ in range(3):
    run()
_lib.lib().wm_debug_bwd_variant(ctypes.c_int(1 << 21))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
dx, part, _ = run()
e1.record()
torch.cuda.synchronize()
nwg = part.shape[0]
t = part.contiguous().view(torch.int64).reshape(nwg, -1)[:, :32].reshape(nwg, 8, 4).double().cpu()
ntiles = B * (H // 8) * (W // 16) / nwg
print(f"bwd_ws8 {'per-sample gradient' if gv else 'premasked'}: {nwg} workgroups x {ntiles:.0f} tiles, launch {e0.elapsed_time(e1) * 1e3:.1f} us (s_memtime: shader cycles)")
for role, ws, names in (("D", range(0, 4), ("MFMA loop + dy staging", "wait at the barrier", "epilogue")), ("W", range(4, 8), ("MFMA loop", "a-tile staging", "wait at the barrier"))):
    out = float(t[:, list(ws), 3].mean())
    m = t[:, list(ws), :3].mean(dim=(0, 1)) / ntiles
    tot = float(m.sum())
    print(f"  {role} waves: " + "; ".join(f"{n} {float(v):.0f} ({100 * float(v) / tot:.0f} %)" for n, v in zip(names, m)) + f"; sum {tot:.0f} cycles/tile; outside the tile loop {out:.0f} cycles per run ({100 * out / (out + tot * ntiles):.1f} % of the wave's time)")
    per = t[:, list(ws), :3].mean(dim=0) / ntiles
    for i, wv in enumerate(ws):
        print(f"     wave {wv}: " + " ".join(f"{float(v):8.0f}" for v in per[i]))
