"""ON THE GPU BOX (round 4 experiment): do two INDEPENDENT chains of the persistent body kernels run faster side by side on half the chip each
than one after the other on the whole chip?  A persistent kernel spends a fixed part of every launch outside its tile loop (filter -> LDS,
first HBM round trip, for the one-pass backward the weight-gradient slab: 11 % of a launch) and the chip idles through every
launch's ramp and tail; with 128-workgroup grids two launches from two streams occupy disjoint CUs (one such workgroup fills a CU's LDS), the
fixed part is paid by half as many workgroups and one chain's ramps / tails sit under the other's steady state.

    python tools/bench_two_chains.py <variant name of a -DWM_MAX_WGS=128 build (tools/build_variant.sh)> [rounds]

Same process, same tensors, interleaved rounds: (a) release library, one stream, 2N launches; (b) variant library, two streams, N launches
each.  Kernels: the forward 64 -> 64 conv (BatchNorm statistics, fused input transform) and the one-pass backward (premasked form)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                                    # noqa: E402
from video_watermarking_forgery_detection_amd import _lib, ops                  # noqa: E402

name = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
full = _lib.lib()
half = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "ab", f"libwm_hip_{name}.so"))
half.wm_last_error_string.restype = ctypes.c_char_p
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
N = 12


def operands():
    g = torch.randn(B, H, W, C, device="cuda").to(dt); y = torch.randn(B, H, W, C, device="cuda").to(dt); xr = torch.randn(B, H, W, C, device="cuda").to(dt)
    stats = torch.rand(4, C, device="cuda") + 0.5; coef = torch.rand(3, C, device="cuda") * 0.01; coef[0] += 1.0
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.3
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    return dict(g=g, y=y, xr=xr, stats=stats, coef=coef, sc=sc, sh=sh, w=w, dw=torch.zeros(C, C, 3, 3, device="cuda"), bias=torch.zeros(C, device="cuda"))


sets = [operands(), operands()]


def use(lib):
    _lib._lib = lib
    for s in sets:
        s["wp"] = ops.pack_w3x3(s["w"], C, C, dt)
        s["wpt"] = ops.pack_w3x3(s["w"], C, C, dt, transpose=True)


def fwd(s):
    ops.conv3x3_fwd(s["xr"], s["wp"], s["bias"], s["sc"], s["sh"], want_stats=True)


def bwd(s):
    ops.conv3x3_bwd_fused(s["g"], s["y"], s["stats"], s["coef"], s["wpt"], s["xr"], s["sc"], s["sh"], s["dw"], False, premasked=True)


def serial(lib, fn):
    use(lib)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        fn(sets[0]); fn(sets[1])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def parallel(lib, fn):
    use(lib)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(s1):
            fn(sets[0])
        with torch.cuda.stream(s2):
            fn(sets[1])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


for label, fn in (("forward 64->64", fwd), ("one-pass backward", bwd)):
    for _ in range(2):
        serial(full, fn); parallel(half, fn); serial(half, fn); parallel(full, fn)
    res = {"full/1 stream": [], "half/2 streams": [], "half/1 stream": [], "full/2 streams": []}
    for r in range(rounds):
        res["full/1 stream"].append(serial(full, fn))
        res["half/2 streams"].append(parallel(half, fn))
        res["half/1 stream"].append(serial(half, fn))
        res["full/2 streams"].append(parallel(full, fn))
    print(label, f"({2 * N} launches per measurement)")
    for k, v in res.items():
        v.sort()
        print(f"    {k:16s} median {v[len(v) // 2]:8.3f} ms   = {1e3 * v[len(v) // 2] / (2 * N):7.1f} us per launch   (min {v[0]:.3f}, max {v[-1]:.3f})")
