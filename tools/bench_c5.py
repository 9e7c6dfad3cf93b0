"""step time of the widened path (SURVEY §8 'next' rows): IRNrhiModel.feed_data / optimize_parameters with the attack cycle
(DiffJPEG, blur, resize, crop, ...), quantisation and the UNet localisation head -- options/train/train_hidden_c5.yml."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
from video_watermarking_forgery_detection_amd.options import options as option
from video_watermarking_forgery_detection_amd.train import synthetic_batches
yml = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "video_watermarking_forgery_detection_amd", "options", "train", "train_hidden_c5.yml")
opt = option.parse(yml, is_train=True)
opt['dist'] = False
torch.manual_seed(10)
model = IRNrhiModel(opt)
B = opt['datasets']['train']['batch_size']
N = 40
batches = list(synthetic_batches(opt, B, 0, N))
step = 0
times = []
for i, data in enumerate(batches):
    step += 1
    model.feed_data(data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    logs, _ = model.optimize_parameters(step, None)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    if i % 8 == 0:
        print(step, [(k, round(v, 4) if isinstance(v, float) else v) for k, v in (logs or [])][:8])
frames = len(model.real_H)
ts = sorted(times[8:])
print(f"frames/step {frames}; step ms median {ts[len(ts)//2]:.2f} min {ts[0]:.2f} max {ts[-1]:.2f} -> {frames/ts[len(ts)//2]*1e3:.0f} frames/s (attack varies per step)")
