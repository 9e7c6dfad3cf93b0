"""diagnostic: where the wave-specialised 64-channel conv kernel spends its cycles (s_memtime stamps per role/phase)
and the shader clock it actually runs at (s_memtime vs the 100 MHz s_memrealtime)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import ops, _lib
_lib._lib = _lib.debug_lib()   # the wm_debug_* switches exist only in the -DWM_DEBUG build (lib/libwm_hip_dbg.so)
B, H, W, C = 16, 256, 256, 64
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wp = ops.pack_w3x3(w, C, C, torch.bfloat16)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
L = _lib.lib()
cn = ["pass A: MFMA half 0 (+ drain of previous half 1)", "pass B: MFMA half 1 (+ drain of half 0)", "barrier wait"]
pn = ["loads tile+2 interleaved with transform/write tile+1", "LDS write drain", "barrier wait"]
MODES = [(0, "full kernel"), (8, "full kernel, producers at default priority"), (1, "no MFMA loop"), (2, "no stores"), (4, "no halo loads"), (3, "no MFMA, no stores"), (5, "no MFMA, no loads"), (6, "no stores, no loads")]
for data in ("normal",):
    x = (torch.randn(B, H, W, C, device="cuda") if data == "normal" else torch.zeros(B, H, W, C, device="cuda")).bfloat16()
    y = torch.empty_like(x)
    for xf, (dbg, dname) in [(x_, m_) for x_ in (True, False) for m_ in MODES]:
        stamps = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
        stat = torch.zeros(256, 2, C, device="cuda")
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        for it in range(12):
            if it == 2: a.record()
            L.wm_debug_conv3x3_ws64_phases(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                           ctypes.c_void_p(sc.data_ptr() if xf else 0), ctypes.c_void_p(sh.data_ptr() if xf else 0),
                                           ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(stat.data_ptr() if xf else 0), B, H, W,
                                           ctypes.c_void_p(stamps.data_ptr()), dbg, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        s = stamps.view(256, 16).double().cpu()
        cyc = s[:, 4].median().item(); rt = s[:, 5].median().item()
        print(f"input={data} xform+stats={xf} [{dname}]: {us:.1f} us/launch; workgroup lifetime median {cyc:.0f} cycles = {rt/100:.1f} us (min {s[:,5].min().item()/100:.1f}, max {s[:,5].max().item()/100:.1f}) -> shader clock {cyc/rt*100:.0f} MHz")
        for role, names, off in (("consumer", cn, 0), ("producer", pn, 8)):
            tot = 0
            for k, nm in enumerate(names):
                v = s[:, off + k].median().item() / 16
                tot += v
                print(f"   {role} {nm:58s} {v:9.0f} cycles/tile")
            print(f"   {role} {'total':58s} {tot:9.0f}")
