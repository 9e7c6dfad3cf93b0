import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import detgen
from oracle import hidden_ref
import video_watermarking_forgery_detection_amd as wm
from video_watermarking_forgery_detection_amd.hidden_models import Decoder
from video_watermarking_forgery_detection_amd import engine, ops
from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
H = 32
x = detgen.uniform((4, 3, H, H), 1); gy = detgen.normal((4, 30), 2)
def ref(dt):
    r = detgen.fill_module(hidden_ref.Decoder(hidden_ref.HiDDenConfiguration(H=H, W=H))).to(dt).train()
    ys = []
    def hook(mod, inp, out):
        out.retain_grad(); ys.append(out)
    for blk in r.layers:
        if hasattr(blk, "layers"):
            blk.layers[0].register_forward_hook(hook)
    out = r(x.to(dt)); (out * gy.to(dt)).sum().backward()
    return r, ys
r64, y64 = ref(torch.float64); r32, y32 = ref(torch.float32)
m = detgen.fill_module(Decoder(HiDDenConfiguration(H=H, W=H))).cuda().train()
wm.set_compute_dtype(m, torch.float32)
out, ctx = m.fwd(x.cuda())
grads = engine.grad_dict(m)
# replicate bwd but capture dy per layer
import types
dys = {}
orig = ops.bn_bwd
def cap(g, gvec, y, stats, C, gamma, dgamma, dbeta, acc, dbias):
    dy = orig(g, gvec, y, stats, C, gamma, dgamma, dbeta, acc, dbias)
    dys[len(dys)] = (dy, y)
    return dy
ops.bn_bwd = cap
m.bwd(ctx, gy.cuda(), grads, accumulate=False, need_input_grad=True)
def err(a, b): return float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
n = len(y64)
for k in range(n):
    li = n - 1 - k
    dy, y = dys[k]
    C = y64[li].shape[1]
    dy_n = dy[..., :C].permute(0, 3, 1, 2); y_n = y[..., :C].permute(0, 3, 1, 2)
    print("layer %d  y: gpu %.2e cpu32 %.2e | dy: gpu %.2e cpu32 %.2e | sum(dy) gpu %.2e" % (li, err(y_n, y64[li]), err(y32[li], y64[li]), err(dy_n, y64[li].grad), err(y32[li].grad, y64[li].grad), float(dy.float().sum().abs() / dy.float().abs().sum())))
for (n1, p), (n2, p64), (n3, p32) in zip(m.named_parameters(), r64.named_parameters(), r32.named_parameters()):
    print("%-30s gpu %.2e cpu32 %.2e" % (n1, err(p.grad, p64.grad), err(p32.grad, p64.grad)))
