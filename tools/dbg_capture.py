"""bisect helper for glayers.CapturedStep: python tools/dbg_capture.py <eager_first 0|1> <mid_term 0|1> <fill 0|1> <side_effects 0|1|2>
side_effects 1: fn rebinds outer names to tensors that require grad (keeps its autograd graph alive: CapturedStep refuses it since round 3;
1 0 0 1 crashed inside hipStreamEndCapture before), 2: the same tensors detached"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import detgen
from video_watermarking_forgery_detection_amd import glayers as G
from video_watermarking_forgery_detection_amd.models.invertible_net import Inveritible_Decolorization_PAMI as PAMI, ResBlock
eager_first, mid_term, fill, side = (int(v) for v in sys.argv[1:5])
plain = int(sys.argv[6]) if len(sys.argv) > 6 else 0   # 1: a plain torch parameter in the loss, whose gradient goes through autograd's AccumulateGrad (the conv weights' no longer do)
impl = int(sys.argv[5]) if len(sys.argv) > 5 else 0   # 0: glayers.CapturedStep; 1: round 2's (two warm-ups on a side stream, capture); 2: that + gc.collect() before the capture


class LegacyCapturedStep:
    def __init__(self, fn, warmup=2, collect=False):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        if collect:
            import gc
            gc.collect()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = fn()

    def replay(self):
        self.graph.replay()
        return self.result
net = PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=torch.bfloat16)
if fill:
    net = detgen.fill_f2(net)
net = net.cuda()
opt = G.FlatAdamW(net, lr=1e-4)
xs = torch.rand(2, 4, 32, 32, device="cuda")
gain = torch.nn.Parameter(torch.ones(4, 1, 1, device="cuda"))
out = {}
def fwd_bwd():
    y = net(xs)
    back, mid = net(y, rev=True)
    loss = ((y - xs) ** 2).mean() + (back ** 2).mean()
    if mid_term:
        loss = loss + (mid ** 2).mean()
    if plain:
        loss = loss + ((y * gain) ** 2).mean()
        gain.grad = None if gain.grad is None else gain.grad.zero_()
    opt.zero_grad()
    loss.backward()
    if side == 1:
        out["y"], out["loss"] = y, loss
    elif side == 2:
        out["y"], out["loss"] = y.detach(), loss.detach()
    return loss
if eager_first:
    fwd_bwd()
try:
    step = G.CapturedStep(fwd_bwd) if impl == 0 else LegacyCapturedStep(fwd_bwd, collect=impl == 2)
except RuntimeError as e:
    print("refused", sys.argv[1:5], str(e)[:90])
    sys.exit(0)
step.replay()
torch.cuda.synchronize()
print("ok", sys.argv[1:], float(step.result.detach()))
