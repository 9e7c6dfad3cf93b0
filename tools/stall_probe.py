"""light instrumentation (two perf_counter reads per call) of every C-ABI call and torch allocation: which one blocks > 2 ms?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import _lib
from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
from video_watermarking_forgery_detection_amd.options import options as option
from video_watermarking_forgery_detection_amd.train import synthetic_batches
yml = os.path.join(ROOT, "video_watermarking_forgery_detection_amd", "options", "train", "train_hidden_c5.yml")
opt = option.parse(yml, is_train=True); opt['dist'] = False
torch.manual_seed(10)
model = IRNrhiModel(opt)
batches = list(synthetic_batches(opt, 1, 0, 40))
slow = []
last = [time.perf_counter(), "start"]
class Wrap:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not callable(f): return f
        def g(*a):
            t0 = time.perf_counter()
            if t0 - last[0] > 0.002: slow.append(("between", last[1], name, round((t0 - last[0]) * 1e3, 1)))
            r = f(*a)
            t1 = time.perf_counter()
            if t1 - t0 > 0.002: slow.append(("inside", name, round((t1 - t0) * 1e3, 1)))
            last[0], last[1] = t1, name
            return r
        if hasattr(f, "restype"):
            try: g.restype = f.restype
            except Exception: pass
        return g
real = _lib.lib()
_lib._lib = Wrap(real)
import torch.utils._python_dispatch  # noqa
steps = []
orig_item = torch.Tensor.item
def timed_item(self):
    t0 = time.perf_counter(); r = orig_item(self); dt = time.perf_counter() - t0
    if dt > 0.002: slow.append(("item", round(dt * 1e3, 1), "after", last[1]))
    last[0] = time.perf_counter()
    return r
torch.Tensor.item = timed_item
orig_tolist = torch.Tensor.tolist
def timed_tolist(self):
    t0 = time.perf_counter(); r = orig_tolist(self); dt = time.perf_counter() - t0
    if dt > 0.002: slow.append(("tolist", round(dt * 1e3, 1), "after", last[1]))
    last[0] = time.perf_counter()
    return r
torch.Tensor.tolist = timed_tolist
for i, data in enumerate(batches):
    model.feed_data(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last[0], last[1] = t0, "step-start"
    model.optimize_parameters(i + 1, None)
    torch.cuda.synchronize()
    steps.append(round((time.perf_counter() - t0) * 1e3, 1))
    if i == 15: slow.clear()
print(steps[16:])
for s in slow[:50]: print(s)
print("n", len(slow))
