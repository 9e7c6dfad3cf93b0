"""ON THE GPU BOX: same-process A/B of library VARIANTS (tools/build_variant.sh) on ONE kernel with fixed random operands -- for timing probes
whose results are wrong by construction (a probe inside the training step would feed its garbage to the next layer and move the clock with it).

    python tools/ab_fwd_kernel.py fwd|bwd <variant> [<variant> ...] [rounds=5]

fwd: the forward 64 -> 64 convolution (fused input transform, BatchNorm statistics); bwd: the one-pass backward (premasked form).
B = 16, 256 x 256, bf16; 40 launches per measurement, the variants interleaved, `rounds` rounds; prints the median per launch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                                    # noqa: E402
from video_watermarking_forgery_detection_amd import _lib, ops                  # noqa: E402

which = sys.argv[1]
names = [a for a in sys.argv[2:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 5
release = _lib.lib()
libs = {}
for n in names:
    if n == "release":
        libs[n] = release
        continue
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "ab", f"libwm_hip_{n}.so"))
    L.wm_last_error_string.restype = ctypes.c_char_p
    libs[n] = L
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
g = torch.randn(B, H, W, C, device="cuda").to(dt); y = torch.randn(B, H, W, C, device="cuda").to(dt); xr = torch.randn(B, H, W, C, device="cuda").to(dt)
stats = torch.rand(4, C, device="cuda") + 0.5; coef = torch.rand(3, C, device="cuda") * 0.01; coef[0] += 1.0
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
dw = torch.zeros(C, C, 3, 3, device="cuda"); bias = torch.zeros(C, device="cuda")
wp = ops.pack_w3x3(w, C, C, dt); wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
N = 40


def run(lib):
    _lib._lib = lib
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    for i in range(N + 3):
        if i == 3:
            a.record()
        if which == "fwd":
            ops.conv3x3_fwd(xr, wp, bias, sc, sh, want_stats=True)
        else:
            ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, sc, sh, dw, False, premasked=True)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / N


res = {n: [] for n in names}
for n in names:
    run(libs[n])
for r in range(rounds):
    for n in names:
        res[n].append(run(libs[n]))
for n in names:
    v = sorted(res[n])
    print(f"{which} {n:12s} median {v[len(v) // 2]:7.2f} us per launch   (min {v[0]:.2f}, max {v[-1]:.2f})")
