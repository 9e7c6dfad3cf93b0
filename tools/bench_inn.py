"""step time of the invertible embedder (row f2: models/invertible_net.py Inveritible_Decolorization_PAMI, the reference's defaults
down_num=3, block_num=[8,8,8], ResBlock subnets) on 4-channel 256x256 frames: embed (forward), extract (rev=True) of the embedded frames,
an L2 loss on both, backward, AdamW -- the use the reference's IRN models make of it (models/IRNrhi_model.py:425-560's generator calls).
usage: python tools/bench_inn.py [bs=8] [dtype=bf16|f16|f32] [steps=10] [graph]   -> one JSON line
(graph: forward + reverse + backward captured once in a hipGraph through torch.cuda.graph and replayed: ~10k launches a step are host-bound otherwise)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_watermarking_forgery_detection_amd import glayers as G
from video_watermarking_forgery_detection_amd.models.invertible_net import Inveritible_Decolorization_PAMI, ResBlock
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
SIZE = int(os.environ.get("INN_SIZE", "256"))
BLOCKS = [int(v) for v in os.environ.get("INN_BLOCKS", "8,8,8").split(",")]
from video_watermarking_forgery_detection_amd.models import invertible_net as _inn
_inn.PARALLEL_SUBNETS = os.environ.get("INN_PAR", "1") == "1"     # a coupling's s / t subnets side by side on two streams
net = Inveritible_Decolorization_PAMI(dims_in=[[4, SIZE, SIZE]], block_num=BLOCKS, subnet_constructor=ResBlock, dtype=dt).cuda()   # the reference's init: every subnet ends in a zero conv
opt = G.FlatAdamW(net, lr=1e-5)
if os.environ.get("INN_PLAN", "1") == "1":
    from video_watermarking_forgery_detection_amd import ops
    G.set_pack_plan(ops.PackPlan())     # packed 3x3 weights: one re-pack launch per optimiser step instead of 2,800 small ones
x = torch.rand(bs, 4, SIZE, SIZE, device="cuda")
GRAPH = len(sys.argv) > 4 and sys.argv[4] == "graph"
def fwd_bwd():
    y = net(x)
    back, mid = net(y, rev=True)
    loss = ((y - x) ** 2).mean() + ((back - x) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    return loss
fwd_bwd(); opt.step()                      # (one eager step: registers the packed weights, the optimiser's refresh validates the plan)
graph = G.CapturedStep(fwd_bwd) if GRAPH else None
if GRAPH:
    loss = graph.result
ms = []
for i in range(N + 2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if graph is not None:
        graph.replay()
    else:
        loss = fwd_bwd()
    opt.step()
    b.record()
    torch.cuda.synchronize()
    if i >= 2:
        ms.append(a.elapsed_time(b))
ms.sort()
try:   # FLOPs as counted on the CPU oracle (tools/count_flops.py), default configuration at 256 x 256 only
    gf = json.load(open(os.path.join(ROOT, "profiles", "r03_f1_f2_flop_counts.json")))["f2_embedder_step"]["gflop_per_frame_256"] * bs if (SIZE == 256 and BLOCKS == [8, 8, 8]) else None
except (OSError, KeyError, ValueError):
    gf = None
_med = ms[len(ms) // 2]
_peak = 157.3 if dt == torch.float32 else 2500.0
print(json.dumps({"step_gflop": gf, "tflops": gf / _med if gf else None, "flops_frac_of_mfma_peak": gf / _med / _peak if gf else None, "graph": GRAPH, "parallel_subnets": _inn.PARALLEL_SUBNETS, "frames_per_step": bs, "dtype": str(dt).split(".")[1], "ms_per_step_median": ms[len(ms) // 2], "ms_per_step_min": ms[0],
                  "frames_per_s": bs / ms[len(ms) // 2] * 1e3, "params": sum(p.numel() for p in net.parameters()), "loss": float(loss.detach())}))
