"""step time of the reference's literal IRNrhi step (models/IRNrhi_literal.py: QF_predictor + FBCNN + Discriminator on the general HIP
layer family) at the reference's size: six quality copies of bs frames, 256x256, default network widths
usage: python tools/bench_literal.py [bs=4] [dtype=bf16|f16|f32] [steps=12]   -> one JSON line"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd.models.IRNrhi_literal import IRNrhiLiteralModel
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 12
torch.manual_seed(3)
model = IRNrhiLiteralModel({"gpu_ids": [0], "is_train": True, "dist": False,
                            "train": {"lr_D": 1e-4, "beta1": 0.9, "beta2": 0.999, "weight_decay_G": 0.0, "gradient_clipping": 1.0, "compute_dtype": dt}})
with torch.no_grad():
    model.localizer.BayarConv2D.weight.uniform_(0.5, 1.5)
base = torch.rand(bs, 3, 256, 256)
imgs = [torch.clamp(base + 0.02 * q * torch.randn(bs, 3, 256, 256), 0, 1) for q in range(6)]
ms = []
for i in range(N + 2):
    model.feed_data((imgs, None))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    logs, _ = model.optimize_parameters(i)
    b.record()
    torch.cuda.synchronize()
    if i >= 2:
        ms.append(a.elapsed_time(b))
ms.sort()
params = {k: sum(p.numel() for p in getattr(model, k).parameters()) for k in ("generator", "localizer", "discriminator")}
# FLOPs of the step as torch.utils.flop_counter counts them on the CPU oracle (tools/count_flops.py -> profiles/r03_f1_f2_flop_counts.json)
try:
    gf = json.load(open(os.path.join(ROOT, "profiles", "r03_f1_f2_flop_counts.json")))["f1_literal_step"]["gflop_per_frame_256"] * 6 * bs
except (OSError, KeyError, ValueError):
    gf = None
med = ms[len(ms) // 2]
peak = 157.3 if dt == "f32" else 2500.0
print(json.dumps({"frames_per_step": 6 * bs, "dtype": dt, "ms_per_step_median": med, "ms_per_step_min": ms[0], "params": params,
                  "step_gflop": gf, "tflops": gf / med if gf else None, "flops_frac_of_mfma_peak": gf / med / peak if gf else None,
                  "logs": {k: round(v, 5) for k, v in logs}}))
