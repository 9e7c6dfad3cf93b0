"""step time of the widened configurations through the model surface (IRNrhiModel.feed_data / optimize_parameters):
  C5 (default yml): 16-frame 256x256 clip folded into the batch, rotating attack set, gradient clipping, UNet localisation head
  C3 (train_hidden_c3.yml): 16 frames, the 7-attack cycle + discriminator, no localiser
usage: python tools/bench_c5.py [yml] [dtype override: bf16|f16|f32] [steps] [deferred]   -> one JSON line (per-attack medians, events on the stream)
`deferred`: train.deferred_logs -- a step's logs are read after the NEXT step has been enqueued (as train.py then feeds its progress bar), so
the host never waits for the GPU between steps; a step's time is then the interval between consecutive steps' end events."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
from video_watermarking_forgery_detection_amd.options import options as option
from video_watermarking_forgery_detection_amd.train import synthetic_batches
D = os.path.join(ROOT, "video_watermarking_forgery_detection_amd", "options", "train")
yml = sys.argv[1] if len(sys.argv) > 1 else os.path.join(D, "train_hidden_c5.yml")
if not os.path.exists(yml):
    yml = os.path.join(D, yml)
opt = option.parse(yml, is_train=True)
opt['dist'] = False
if len(sys.argv) > 2:
    opt['train']['compute_dtype'] = sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
deferred = "deferred" in sys.argv[4:]
if deferred:
    opt['train']['deferred_logs'] = True
torch.manual_seed(10)
model = IRNrhiModel(opt)
B = opt['datasets']['train']['batch_size']
per = {}
# the synthetic batches are drawn BEFORE the timed loop: torch.rand on the host runs on every core it sees, and on a box whose CPU
# share is a cgroup quota (16 CPUs of 256 here) that burst exhausts the 100 ms quota period -- the whole process, the thread that
# feeds the GPU included, is then throttled for 40-60 ms every few steps (measured: tools/stall_probe*.py; a real loader's workers
# are separate processes)
ends, kinds, pend = {}, {}, {}   # deferred: end event, attack name and unread logs by step
# pinned, as the training loader hands them over (data/__init__.py): feed_data's copy is then asynchronous
batches = [tuple(t.pin_memory() for t in d) for d in synthetic_batches(opt, B, 0, N)]
for i, data in enumerate(batches):
    step = i + 1
    if step == 31:
        torch.cuda.synchronize()
        wall0 = time.perf_counter()
    model.feed_data(data)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    logs, _ = model.optimize_parameters(step, None)
    b.record()
    if deferred:
        ends[step], kinds[step], pend[step] = b, model.attack.name, logs
        q = step - 1
        if q in pend:
            lq = pend.pop(q)
            n = len(lq)      # waits for step q's scalars; step q + 1 is already enqueued behind it
            if n and q > 30 and (q - 1) in ends:   # a step's time in a loop that never drains the queue: end of the step before -> its own end
                ends[q].synchronize()
                per.setdefault(dict(lq).get('Kind', kinds[q]), []).append(ends[q - 1].elapsed_time(ends[q]))
        continue
    torch.cuda.synchronize()
    if logs and step > 30:     # every attack has run four times by then: first-use allocations and (train.graph, configuration C3) the capture of
                               # each layer's step -- its third call -- are out of the way
        d = dict(logs)
        per.setdefault(d.get('Kind', model.attack.name), []).append(a.elapsed_time(b))
torch.cuda.synchronize()
wall_ms = 1e3 * (time.perf_counter() - wall0) / (N - 30)   # the whole loop, feed_data included: what a training run sees
frames = len(model.real_H)
med = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
allt = sorted(t for v in per.values() for t in v)
out = {"yml": os.path.basename(yml), "dtype": opt['train']['compute_dtype'], "frames_per_step": frames, "steps_timed": len(allt),
       "ms_per_step_median": allt[len(allt) // 2], "ms_per_step_mean": sum(allt) / len(allt), "ms_per_step_max": allt[-1],
       "frames_per_s": frames / (sum(allt) / len(allt)) * 1e3, "ms_per_step_by_attack": {k: round(v, 3) for k, v in med.items()},
       "amp_scale": model.amp.get_scale() if model.amp is not None else None, "deferred_logs": deferred, "wall_ms_per_step": wall_ms, "wall_frames_per_s": frames / wall_ms * 1e3, "graph": model.hidden._graphs is not None}
print(json.dumps(out))
