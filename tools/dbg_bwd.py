"""Debugging aid for csrc/bwd_ws.hip: the one-kernel backward against the two-kernel form on small ragged shapes -- where the two differ,
which side is off against an f64 evaluation, and (with an identity filter, so that dx == dy) the staged dy itself.
usage: python tools/dbg_bwd.py   (on the GPU box)"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import detgen
from video_watermarking_forgery_detection_amd import ops


def nhwc(x, dtype, ld=None):
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, ld or C, dtype=dtype, device="cuda")
    out[..., :C] = x.permute(0, 2, 3, 1).to(dtype).cuda()
    return out


for (B, H, W, dt, rev) in ((1, 21, 37, torch.float16, False), (1, 21, 37, torch.bfloat16, False), (1, 24, 48, torch.float16, False)):
    C = 64
    g = nhwc(detgen.normal((B, C, H, W), 901), dt, C); y = nhwc(detgen.normal((B, C, H, W), 902, mean=0.2), dt, C)
    xr = nhwc(detgen.normal((B, C, H, W), 903, mean=0.1), dt, C)
    stats = torch.empty(4, C, device="cuda")
    stats[0] = detgen.normal((C,), 904, mean=1.0, std=0.3).cuda(); stats[1] = detgen.normal((C,), 905, std=0.3).cuda()
    stats[2] = detgen.normal((C,), 906, std=0.2).cuda(); stats[3] = detgen.uniform((C,), 907).cuda() + 0.5
    coef = torch.empty(3, C, device="cuda")
    coef[0] = detgen.normal((C,), 908, mean=1.0, std=0.2).cuda(); coef[1] = detgen.normal((C,), 909, std=0.01).cuda(); coef[2] = detgen.normal((C,), 910, std=0.01).cuda()
    in_scale = detgen.normal((C,), 911, mean=1.0, std=0.3).cuda(); in_shift = detgen.normal((C,), 912, std=0.3).cuda()
    w = detgen.normal((C, C, 3, 3), 913, std=0.05).cuda()
    wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
    dw0 = torch.zeros(C, C, 3, 3, device="cuda"); dw1 = torch.zeros(C, C, 3, 3, device="cuda")
    dy, dx0, part0 = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, xr, in_scale, in_shift, reverse=rev)
    ops.conv3x3_wgrad(xr, C, in_scale, in_shift, dy, dw0, False, reverse=not rev)
    dx1, part1, _ = ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, in_scale, in_shift, dw1, False, reverse=rev)
    d = (dx0.float() - dx1.float()).abs()
    idx = torch.nonzero(d > 0)
    print(dt, H, W, "ndiff", idx.shape[0], "max", d.max().item(), "first", idx[:6].tolist(), "rows", sorted(set(idx[:, 1].tolist()))[:30],
          "cols", sorted(set(idx[:, 2].tolist()))[:40])
    print("  dw rel", ((dw0 - dw1).abs().max() / dw0.abs().max()).item(), "sums", (part0.double().sum(0) - part1.double().sum(0)).abs().max().item())

# which side is odd?  dy (written by the two-kernel form) against an f64 evaluation of the apply pass, per pixel
B, H, W, dt = 1, 21, 37, torch.float16
C = 64
g = nhwc(detgen.normal((B, C, H, W), 901), dt, C); y = nhwc(detgen.normal((B, C, H, W), 902, mean=0.2), dt, C)
xr = nhwc(detgen.normal((B, C, H, W), 903, mean=0.1), dt, C)
gd, yd = g.double(), y.double()
sc, sh, mean, inv = [stats[i].double() for i in range(4)]
ca, c1, c2 = [coef[i].double() for i in range(3)]
z = sc * yd + sh
ref = ca * (gd * (z > 0) - c1 - (yd - mean) * inv * c2)
dy, dx0, part0 = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, xr, in_scale, in_shift)
err = (dy.double() - ref).abs() / (ref.abs() + 1e-3)
print("dy vs f64: max rel", err.max().item(), "per-pixel max at (0,8):", err[0, 0, 8].max().item(), "median pixel max", err.amax(-1).median().item())
worst = torch.nonzero(err.amax(-1) > 3 * err.amax(-1).median())
print("pixels with outlying dy error:", worst[:10].tolist())
for rep in range(3):
    dxa, _, _ = ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, in_scale, in_shift, torch.zeros(C, C, 3, 3, device="cuda"), False)
    _, dxb, _ = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, xr, in_scale, in_shift)
    print("rep", rep, "fused vs first fused equal:", torch.equal(dxa, dx1) if rep else None, "two-kernel vs its first:", torch.equal(dxb, dx0),
          "ndiff fused/two-kernel", int((dxa != dxb).sum()))
    if rep == 0:
        dx1 = dxa
# f64 reference of dx at the differing spots
import torch.nn.functional as F
dyh = dy.double().permute(0, 3, 1, 2)
wd = w.to(dt).double()
dx_ref = F.conv_transpose2d(dyh, wd, padding=1).permute(0, 2, 3, 1)
mask = dxa != dxb
print("at differing spots: |fused - ref|", (dxa.double() - dx_ref)[mask].abs().mean().item(), "|two-kernel - ref|", (dxb.double() - dx_ref)[mask].abs().mean().item())

# identity filter: dx == dy, so the fused kernel's staged dy becomes visible
wi = torch.zeros(C, C, 3, 3, device="cuda")
for c in range(C):
    wi[c, c, 1, 1] = 1.0
wpi = ops.pack_w3x3(wi, C, C, dt, transpose=True)
dyo, dxo, _ = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpi, xr, in_scale, in_shift)
dxf, _, _ = ops.conv3x3_bwd_fused(g, y, stats, coef, wpi, xr, in_scale, in_shift, torch.zeros(C, C, 3, 3, device="cuda"), False)
print("two-kernel: dx == dy", torch.equal(dyo, dxo), " fused dx == dy", torch.equal(dyo, dxf))
idx = torch.nonzero(dyo != dxf)
print("differing dy elements", idx.tolist()[:10])
for (b_, r_, c_, ch) in idx.tolist()[:4]:
    gv, yv = g[b_, r_, c_, ch].item(), y[b_, r_, c_, ch].item()
    print("  g", gv, "y", yv, "dy two-kernel", dyo[b_, r_, c_, ch].item(), "fused", dxf[b_, r_, c_, ch].item(), "f64", ref[b_, r_, c_, ch].item(),
          "scale", stats[0, ch].item(), "shift", stats[1, ch].item(), "z", (stats[0, ch] * yv + stats[1, ch]).item())
