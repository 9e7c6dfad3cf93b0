"""ON THE GPU BOX: does a kernel's result change while ANOTHER kernel of the library runs beside it on a second stream?  Fixed operands; the
solo result is the reference.  Round 4: the JPEG kernels beside bwd_ws16 / the 16-channel weight gradient -- 22-30 of 30 launches wrong while
those could share a CU, 0 of 180 since they request LDS up to 137,472 B (csrc/wgrad_ws.hip WM_LDS_PAD16).  A library variant built from
the sources before that change (tools/build_variant.sh, WM_LIB_VARIANT) shows the old behaviour."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_watermarking_forgery_detection_amd import ops
from video_watermarking_forgery_detection_amd import noise_layers as NL
dev = "cuda"
B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
torch.manual_seed(0)
x = torch.rand(B, 3, H, W, device=dev)
layer = NL.JpegSS(50)
g = torch.randn(B, H, W, C, device=dev).to(dt); y = torch.randn(B, H, W, C, device=dev).to(dt)
stats = torch.rand(4, C, device=dev) + 0.5; coef = torch.rand(3, C, device=dev) * 0.01; coef[0] += 1.0
x16 = torch.randn(B, H, W, 16, device=dev).to(dt); x16[..., 3:] = 0
w16 = torch.randn(C, 3, 3, 3, device=dev) * 0.05; dw16 = torch.zeros(C, 3, 3, 3, device=dev)
wpt16 = ops.pack_w3x3(w16, C, 16, dt, transpose=True)
w32 = torch.randn(32, 32, 3, 3, device=dev) * 0.05
x32 = torch.randn(B, H, W, 32, device=dev).to(dt)
wp32 = ops.pack_w3x3(w32, 32, 32, dt)
dwf = torch.zeros(C, 16, 3, 3, device=dev)
img = torch.rand(B, 3, H, W, device=dev)
wcat = torch.randn(64, 97, 3, 3, device=dev) * 0.05
msg = torch.randint(0, 2, (B, 30), device=dev).float()
def heavy(kind):
    if kind == "bwd_ws16":
        ops.conv3x3_bwd_fused16(g, y, stats, coef, wpt16, x16, dw16, False)
    elif kind == "generic conv 32->32":
        ops.conv3x3_fwd(x32, wp32, None, None, None, want_stats=False)
    elif kind == "wgrad_ws16<16>":
        ops.conv3x3_wgrad(x16, 16, None, None, g, dwf, False)
    elif kind == "concat_side":
        ops.concat_side_fwd(img, wcat, None, msg, dt, 0, 30, 94)
    elif kind == "jpeg_fwd (itself)":
        ops.jpeg_fwd(img, layer._mode, layer._tables, 0)
    elif kind == "diffjpeg":
        ops.diffjpeg_fwd(img, 1, 1.0) if hasattr(ops, "diffjpeg_fwd") else None
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
L = NL.JpegMask(50)
sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.3
xr = torch.randn(B, H, W, C, device=dev).to(dt)
w64 = torch.randn(C, C, 3, 3, device=dev) * 0.05
wp64 = ops.pack_w3x3(w64, C, C, dt)
gy3 = torch.randn(B, 3, H, W, device=dev)
g16 = torch.randn(B, H, W, 16, device=dev).to(dt)
flat = torch.randn(1 << 20, device=dev); fg = torch.randn(1 << 20, device=dev)
hw = torch.randn(3, 64, device=dev) * 0.1; hb = torch.zeros(3, device=dev)
victims = {
    "jpeg_fwd (mask)": lambda: ops.jpeg_fwd(x, L._mode, L._tables, 0),
    "jpeg_bwd (SS)": lambda: ops.jpeg_bwd(x, gy3, NL.JpegSS(50)._mode, NL.JpegSS(50)._tables, 0),
    "diffjpeg_fwd": lambda: ops.diffjpeg_fwd(x, 1, 1.0),
    "resample_bwd bicubic (LDS x pass)": lambda: ops.resample_bwd(gy3[:, :, :179, :179].contiguous(), None, (256, 256), (0, 256, 0, 256), ops.BICUBIC),
    "median_bwd4": lambda: ops.median_bwd(gy3, torch.zeros(B, 3, H, W, dtype=torch.int8, device=dev), 3),
    "image_grad_mse": lambda: ops.image_grad_mse(g16, x, img, 1e-3)[0],
    "conv1x1 head fwd": lambda: (lambda r: r if isinstance(r, torch.Tensor) else r[0])(ops.conv1x1_head_fwd(xr, sc, sh, hw, hb)),
    "first-layer fwd <16,64> + stats": lambda: torch.cat([t.float().reshape(-1) for t in ops.conv3x3_fwd(x16, ops.pack_w3x3(w16, 64, 16, dt), None, None, None, want_stats=True)]),
}
for name, fn in victims.items():
    try:
        solo = fn(); torch.cuda.synchronize()
    except Exception as e:
        print(name, "skipped:", type(e).__name__, e); continue
    bad = 0
    for it in range(30):
        sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sA):
            heavy("bwd_ws16"); heavy("wgrad_ws16<16>"); heavy("bwd_ws16")
        with torch.cuda.stream(sB):
            out = fn()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, solo))
    print(f"{name:24s} beside bwd_ws16: {bad} / 30 mismatching launches", flush=True)
