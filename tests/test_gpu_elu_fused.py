"""GPU: conv + bias + ELU as one launch and its backward in two (+ the slab reduction) -- rows f1 / f2, the coupling subnets' layers
(/root/reference/models/invertible_net.py:326-366: ResBlock's conv1..conv4 followed by nn.ELU).

  * wm_conv3x3_fwd_elu against torch's fp32 conv2d + elu of the same 16-bit operands (the fused epilogue applies ELU to the f32
    accumulator: at most one rounding of the result away);
  * wm_conv3x3_dgrad_elufused: gz bit for bit against g * (out > 0 ? 1 : out + 1) in fp32 rounded once; dx bit for bit against the plain
    input-gradient kernel on that gz; the bias partial rows against a float64 column sum;
  * wm_conv3x3_wgrad_bias: dw bit for bit against wm_conv3x3_wgrad, db against float64 -- also accumulating;
  * the autograd node (glayers.ConvAct) fused against the separate launches, forward and every gradient, incl. accumulation into a
    FlatAdamW buffer; shapes with partial tiles and every input-channel count the persistent kernel takes."""
import pytest
import torch
import torch.nn.functional as F

import detgen

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, H, W, Cin, seed):
    x = detgen.normal((B, H, W, Cin), seed, std=1.0).to(DEV)
    w = (detgen.normal((64, Cin, 3, 3), seed + 1, std=0.08)).to(DEV)
    b = detgen.normal((64,), seed + 2, std=0.3).to(DEV)
    g = detgen.normal((B, H, W, 64), seed + 3, std=1.0).to(DEV)
    return x, w, b, g


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 32, 32, 64), (1, 24, 40, 64), (2, 16, 48, 32), (3, 20, 20, 16), (8, 64, 64, 64)])
def test_fwd_elu_and_backward_kernels(dtype, shape):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin = shape
    x, w, b, g = _case(B, H, W, Cin, 11 + Cin + H)
    xh, gh = x.to(dtype), g.to(dtype)
    assert ops.conv3x3_fwd_elu_supported(Cin, 64, dtype)
    wp = ops.pack_w3x3(w, 64, Cin, dtype)
    out = ops.conv3x3_fwd_elu(xh, wp, b)
    wq = wp.float().permute(1, 2, 0).reshape(64, Cin, 3, 3)          # the packed (rounded) filter back in conv2d's layout
    ref = F.elu(F.conv2d(xh.float().permute(0, 3, 1, 2), wq, b, padding=1)).permute(0, 2, 3, 1)
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    err = (out.float() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) < 1.5 * ulp, float((err / (ref.abs() + 1.0)).max())
    # the unfused pair (conv stored as 16 bits, then ELU) is at most one more rounding away
    z, _ = ops.conv3x3_fwd(xh, wp, b, None, None, want_stats=False)
    un = ops.unary_fwd(z, "elu")
    assert float(((out.float() - un.float()).abs() / (ref.abs() + 1.0)).max()) < 3 * ulp
    # ---- backward, input-gradient half (16-channel inputs: the filter packed to 32 rows, dx keeps the 16-channel stride)
    assert ops.conv3x3_dgrad_elufused_supported(Cin, dtype)
    wt = ops.pack_w3x3(w, 64, max(Cin, 32), dtype, transpose=True)
    dx, gz, part = ops.conv3x3_dgrad_elufused(gh, out, wt, dx_stride=Cin)
    assert dx.shape == (B, H, W, Cin)
    o32 = out.float()
    gz_ref32 = gh.float() * torch.where(o32 > 0, torch.ones_like(o32), o32 + 1.0)
    assert torch.equal(gz, gz_ref32.to(dtype))
    dx_ref, _ = ops.conv3x3_fwd(gz, wt, None, None, None, want_stats=False)
    assert torch.equal(dx, dx_ref[..., :Cin])
    col = gz_ref32.double().sum(dim=(0, 1, 2))
    got = part.double().sum(dim=0)
    assert float((got - col).abs().max() / (col.abs().max() + 1e-9)) < 1e-5
    dx2, gz2, _ = ops.conv3x3_dgrad_elufused(gh, out, wt, want_gz=False, dx_stride=Cin)
    assert gz2 is None and torch.equal(dx2, dx)
    # ---- weight gradient + the bias gradient in its reduction launch
    dw_ref = torch.empty(64, Cin, 3, 3, device=DEV)
    ops.conv3x3_wgrad(xh, Cin, None, None, gz, dw_ref, False)
    dw, db = torch.empty_like(dw_ref), torch.empty(64, device=DEV)
    ops.conv3x3_wgrad_bias(xh, gz, dw, False, part, db, False)
    assert torch.equal(dw, dw_ref)
    assert float((db.double() - col).abs().max() / (col.abs().max() + 1e-9)) < 1e-5
    dw0, db0 = detgen.normal((64, Cin, 3, 3), 5).to(DEV), detgen.normal((64,), 6).to(DEV)
    dw1, db1 = dw0.clone(), db0.clone()
    ops.conv3x3_wgrad_bias(xh, gz, dw1, True, part, db1, True)
    assert torch.allclose(dw1, dw0 + dw, rtol=0, atol=1e-5 * float(dw.abs().max())) and torch.allclose(db1, db0 + db, rtol=0, atol=1e-5 * float(db.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin", [8, 32, 64])
def test_conv_elu_autograd_node_fused_vs_separate(dtype, cin):
    """glayers.ConvAct(conv, "elu"): the fused node against the separate launches (FUSE_ELU off) -- output, input gradient, weight and bias
    gradients -- and against torch autograd in fp32 at the 16-bit bound; gradients into .grad and into a FlatAdamW buffer."""
    from video_watermarking_forgery_detection_amd import glayers as G
    torch.manual_seed(3)
    B, H, W = 2, 24, 32
    x = detgen.normal((B, cin, H, W), 21, std=1.0)
    gy = detgen.normal((B, 64, H, W), 22, std=1.0)
    layer = G.ConvAct(G.Conv2d(cin, 64, 3, 1, 1), "elu").to(DEV)
    with torch.no_grad():
        layer[0].weight.copy_(detgen.normal((64, cin, 3, 3), 23, std=0.08))
        layer[0].bias.copy_(detgen.normal((64,), 24, std=0.3))

    def run(fused, flat):
        G.FUSE_ELU = fused
        try:
            for p in layer.parameters():
                p.grad = None
            opt = G.FlatAdamW(layer, lr=1e-3) if flat else None
            if opt is not None:
                opt.zero_grad()
            xd = x.to(DEV).requires_grad_(True)
            y = G.to_nchw(layer(G.to_nhwc(xd, dtype)), 64)
            (y * gy.to(DEV)).sum().backward()
            return y.detach(), xd.grad.clone(), layer[0].weight.grad.clone(), layer[0].bias.grad.clone()
        finally:
            G.FUSE_ELU = True

    fused, sep = run(True, False), run(False, False)
    fused_flat = run(True, True)
    # torch fp32 autograd on the 16-bit-rounded operands
    xr = x.to(dtype).float().requires_grad_(True)
    wr = layer[0].weight.detach().cpu().to(dtype).float().requires_grad_(True)
    br = layer[0].bias.detach().cpu().clone().requires_grad_(True)
    yr = F.elu(F.conv2d(xr, wr, br, padding=1))
    (yr * gy).sum().backward()
    refs = (yr.detach(), xr.grad, wr.grad, br.grad)
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    for name, a, b_, r in zip(("out", "gx", "gw", "gb"), fused, sep, refs):
        scale = float(r.abs().max())
        assert float((a.cpu().float() - r).abs().max()) < tol * scale, (name, "fused vs torch")
        assert float((a.float() - b_.float()).abs().max()) < tol * scale, (name, "fused vs separate")
    for name, a, b_ in zip(("out", "gx", "gw", "gb"), fused, fused_flat):
        assert torch.equal(a, b_) if name in ("out", "gx") else torch.allclose(a, b_, rtol=0, atol=1e-6 * float(a.abs().max())), name
