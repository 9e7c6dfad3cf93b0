"""A/B of kernel variants on the REAL training step, interleaved rounds in ONE process on ONE device
(timings from different devices / processes are not comparable: DVFS).
usage: python tools/ab_step.py <debug setter symbol> <variant ids...>   e.g.  wm_debug_ws_variant 0 1"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import _lib, ops
_lib._lib = _lib.debug_lib()   # the wm_debug_* switches exist only in the -DWM_DEBUG build (lib/libwm_hip_dbg.so)
from video_watermarking_forgery_detection_amd.hidden_models import Hidden
from video_watermarking_forgery_detection_amd import noise_layers as NL
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration

if sys.argv[1].startswith("attr:"):   # python attribute of the Hidden object instead of a library knob, e.g. attr:overlap_streams
    _name = sys.argv[1][5:]
    def setter(v):
        setattr(h, _name, bool(v))
else:
    setter = getattr(_lib.lib(), sys.argv[1])
variants = [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda", 0)
torch.manual_seed(10)
h = Hidden(HiDDenConfiguration(H=256, W=256), dev, NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
images = torch.rand(16, 3, 256, 256, device=dev)
messages = torch.randint(0, 2, (16, 30), device=dev).float()
ROUNDS, STEPS = 6, 10
res = {v: [] for v in variants}
kres = {v: [] for v in variants}
for v in variants:
    setter(v)
    for _ in range(3):
        h.train_on_batch([images, messages])
for rnd in range(ROUNDS):
    for v in variants:
        setter(v)
        h.train_on_batch([images, messages])
        tname = os.environ.get("AB_TIME")   # time another op family's launches instead (e.g. AB_TIME=conv3x3_bwd_fused)
        timer = ops.KernelTimer((lambda name, i: name == tname) if tname else (lambda name, i: name == "conv3x3_fwd" and i["Cin"] == 64 and i["CoutP"] == 64))
        ops.set_kernel_timer(timer)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(STEPS):
            h.train_on_batch([images, messages])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / STEPS * 1e3
        ops.set_kernel_timer(None)
        k = timer.elapsed_ms() or [0.0]
        res[v].append(dt); kres[v].append(1e3 * sum(k) / len(k))
for v in variants:
    print(f"variant {v}: step ms median {statistics.median(res[v]):.3f} min {min(res[v]):.3f} | timed launches avg us median {statistics.median(kres[v]):.1f} min {min(kres[v]):.1f} | rounds {['%.2f' % x for x in res[v]]}")
