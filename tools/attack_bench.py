"""Every attack kernel forward + backward at the benchmark size (B=16, 3x256x256 f32 NCHW), each timed with events on the launch
stream -> GB/s of ALGORITHMIC bytes (24 B/px forward = 12 read + 12 written; backward per kernel, see `BYTES`) against
the 8 TB/s HBM peak.  Run under rocprofv3 (--kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE) for the committed summaries.
usage: python tools/attack_bench.py [reps] [size] [batch]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import noise_layers as NL
from video_watermarking_forgery_detection_amd.utils.JPEG import DiffJPEG
from video_watermarking_forgery_detection_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
torch.manual_seed(10)
x = torch.rand(B, 3, S, S, device=dev)
g = torch.randn(B, 3, S, S, device=dev)
px = float(B * S * S)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record()
        fn()
    ev[reps].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


class Fixed:
    def __init__(self, layer, **kw):
        self.layer, self.kw = layer, kw

    def fwd(self, t):
        return self.layer.fwd(t, **self.kw)

    def bwd(self, c, gg):
        return self.layer.bwd(c, gg)


H8 = S // 8
# name -> (layer, forward bytes / px, backward bytes / px): 3 channels x 4 B per tensor pass
cases = {
    "Jpeg50": (NL.Jpeg(50), 24, 12),            # backward writes zeros only (torch.round has zero gradient)
    "JpegSS50": (NL.JpegSS(50), 24, 36),        # backward reads x and g, writes gx
    "JpegMask50": (NL.JpegMask(50), 24, 24),
    "GaussianBlur": (NL.GaussianBlur(), 24, 24),
    "MiddleBlur3": (NL.MiddleBlur(3), 24 + 3, 24 + 3),   # + the int8 argmedian index plane
    "MiddleBlur5": (NL.MiddleBlur(5), 24 + 3, 24 + 3),
    "Resize0.7": (Fixed(NL.Resize(), resize_ratio=0.7), 12 + 2 * 12 * 0.49 + 12, 12 + 12 + 2 * 12 * 0.49 + 12),   # two resample passes through the 0.7x image
    "Crop0.75": (Fixed(NL.Crop(), apex=(H8, H8 + int(0.75 * S), H8, H8 + int(0.75 * S))), 12 * 0.5625 + 12, 12 + 12),
    "Quantization": (None, 24, 0),
}
out = {}
for name, (layer, fb, bb) in cases.items():
    if layer is None:
        ms = timed(lambda: ops.quant(x))
        out[name] = {"fwd_ms": ms, "fwd_GBps": fb * px / (ms * 1e-3) / 1e9, "fwd_frac_of_8TBps": fb * px / (ms * 1e-3) / 8e12}
        continue
    y, c = layer.fwd(x)
    fms = timed(lambda: layer.fwd(x))
    bms = timed(lambda: layer.bwd(c, g))
    out[name] = {"fwd_ms": fms, "fwd_GBps": fb * px / (fms * 1e-3) / 1e9, "fwd_frac_of_8TBps": fb * px / (fms * 1e-3) / 8e12,
                 "bwd_ms": bms, "bwd_GBps": bb * px / (bms * 1e-3) / 1e9, "bwd_frac_of_8TBps": bb * px / (bms * 1e-3) / 8e12,
                 "fwd_bytes_per_px": fb, "bwd_bytes_per_px": bb}
dj = DiffJPEG(True, S, S, 75)    # utils/JPEG.py:501-540 (quality 75, round_only_at_0)
yj, cj = dj.fwd(x)
fms = timed(lambda: dj.fwd(x))
bms = timed(lambda: dj.bwd(cj, g))
out["DiffJPEG75"] = {"fwd_ms": fms, "fwd_GBps": 24 * px / (fms * 1e-3) / 1e9, "fwd_frac_of_8TBps": 24 * px / (fms * 1e-3) / 8e12,
                     "bwd_ms": bms, "bwd_GBps": 36 * px / (bms * 1e-3) / 1e9, "bwd_frac_of_8TBps": 36 * px / (bms * 1e-3) / 8e12,
                     "fwd_bytes_per_px": 24, "bwd_bytes_per_px": 36}
print(json.dumps({"size": S, "batch": B, "reps": reps, "kernels": out}, indent=1))
