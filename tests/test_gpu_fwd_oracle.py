"""Round 4: the two kernels of the benchmarked step that round 3 pinned only through chains of this library's own kernels, DIRECTLY against
fp64 on the host, at the benchmark's full size (B = 16, 256 x 256) in both 16-bit dtypes -- the way tests/test_gpu_bwd_oracle.py does it for
the dominant kernel:

  * the forward 64 -> 64 convolution `conv3x3_ws_kernel<64,64,XFORM,STATS,M16>` in its whole-tile form (csrc/conv3x3_ws.hip; 15 launches of
    the step): Conv3x3(pad 1) of ReLU(BatchNorm(x)) (/root/reference/hidden_models/conv_bn_relu.py:11-15, the BatchNorm + ReLU of the
    block BELOW fused into the staging) + bias, and the BatchNorm statistics sum(y), sum(y^2) of its epilogue;
  * the one-pass backward of an image-fed first layer `bwd_ws16_kernel` (csrc/bwd_ws16.hip): dx (wrt the 16-channel image tensor) and dW.

No other kernel of the library is in either chain: every operand is derived in fp64 from the same 16-bit-rounded tensors the kernel reads.
What a kernel adds to the exact result: the activated input (forward) resp. dy (backward) rounded to 16 bits when staged -- reproduced
exactly in the reference --, f32 accumulation, and the 16-bit storage rounding of the result; the bounds below are those, per quantity.
"""
import pytest
import torch
import torch.nn.functional as F

import detgen

pytestmark = pytest.mark.gpu

C = 64


def _nhwc(x, dt, cp=None):
    """[B,C,H,W] (values representable in dt) -> contiguous NHWC dt tensor on the GPU, channels zero-padded to cp"""
    t = x.permute(0, 2, 3, 1).contiguous().to(dt)
    if cp is not None and cp > t.shape[-1]:
        t = F.pad(t, (0, cp - t.shape[-1]))
    return t.cuda()


def _nchw64(t):
    return t.double().cpu().permute(0, 3, 1, 2)


def _u(dt):
    return 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11     # unit round-off: half an ulp at 1.0


@pytest.mark.parametrize("case", [(16, 256, 256, torch.bfloat16), (16, 256, 256, torch.float16), (2, 32, 48, torch.bfloat16)])
def test_forward_conv_ws_against_fp64(case):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, dt = case
    q = lambda t: t.to(dt).double()                                              # noqa: E731  round to the storage type
    seed = 5100
    x = q(detgen.normal((B, C, H, W), seed + 1, mean=0.1))                       # the block below's raw conv output, as stored
    in_scale = detgen.normal((C,), seed + 2, mean=1.0, std=0.3)                  # its BatchNorm scale / shift (f32 device constants)
    in_shift = detgen.normal((C,), seed + 3, std=0.3)
    # the kernel's staging: one f32 fma, ReLU, rounded to the storage type.  (scale * x is exact in fp64 -- 24 x 11 significant bits --, the
    # sum rounds once to 53 bits and once more to f32: an f32 fma up to double-rounding cases of measure ~2^-29)
    z32 = (in_scale.double().view(1, C, 1, 1) * x + in_shift.double().view(1, C, 1, 1)).float()
    a = q(torch.relu(z32))
    w = q(detgen.normal((C, C, 3, 3), seed + 4, std=(2.0 / (9 * C)) ** 0.5))     # the filter as packed (16 bit)
    bias = detgen.normal((C,), seed + 5, std=0.1)
    y64 = F.conv2d(a, w, bias.double(), padding=1)                               # exact up to fp64 accumulation
    wp = ops.pack_w3x3(w.float().cuda(), C, C, dt)
    y, part = ops.conv3x3_fwd(_nhwc(x, dt), wp, bias.cuda(), in_scale.cuda(), in_shift.cuda(), want_stats=True)
    torch.cuda.synchronize()
    got = _nchw64(y)
    u = _u(dt)
    ymax = y64.abs().max().item()
    # ReLU decisions within f32 round-off of zero are the one place the staged operand may differ (a ~1e-6 fraction of the inputs; the value
    # that flips is itself ~1e-7, so its effect on y is far below the bound -- no exclusion needed)
    err = (got - y64).abs()
    tight = 1.02 * u * y64.abs() + 4e-6 * ymax                                   # correctly rounded + f32 accumulation of 576 products
    rel_l2 = (err.pow(2).sum() / y64.pow(2).sum()).sqrt().item()
    print(f"[conv3x3_ws fwd vs fp64 {dt} {B}x{H}x{W}] y: relative L2 {rel_l2:.3e} = {rel_l2 / u:.3f} u, max |err| / bound {(err / tight).max().item():.3f}")
    assert (err <= tight).all(), (err / tight).max().item()
    assert rel_l2 < 0.5 * u                                                      # one rounding: R u with R ~ 0.43 (tests/test_gpu_bwd_oracle.py)
    # ---- BatchNorm statistics: sums of the f32 accumulator values (before the storage rounding), per channel, to f32 accumulation
    s = part.double().sum(0).cpu()
    assert s.shape == (2, C)
    for k, (ref, l1, name) in enumerate(((y64.sum((0, 2, 3)), y64.abs().sum((0, 2, 3)), "sum y"),
                                         (y64.pow(2).sum((0, 2, 3)), y64.pow(2).sum((0, 2, 3)), "sum y^2"))):
        d = (s[k] - ref).abs()
        print(f"    {name}: max |diff| = {(d / l1).max().item():.2e} of sum|.|")
        assert (d <= 2e-6 * l1).all(), (name, (d / l1).max().item())
    # the mean / variance BatchNorm derives from them (what the step consumes): 1e-5 of the standard deviation
    n = B * H * W
    mean = s[0] / n; var = s[1] / n - mean * mean
    mean64 = y64.mean((0, 2, 3)); var64 = y64.var((0, 2, 3), unbiased=False)
    assert ((mean - mean64).abs() <= 1e-5 * var64.sqrt()).all() and ((var - var64).abs() <= 1e-4 * var64).all()


def _first_layer_operands(B, H, W, dt, seed, premasked):
    """fp64 graph of an image-fed block Conv3x3(3 -> 64) -> BatchNorm2d(train) -> ReLU and the operands wm_conv3x3_bwd_fused16 takes"""
    q = lambda t: t.to(dt).double()                                              # noqa: E731
    x = q(detgen.uniform((B, 3, H, W), seed + 1)).requires_grad_(True)           # the image, as the NHWC16 conversion stores it
    w = q(detgen.normal((C, 3, 3, 3), seed + 2, std=(2.0 / 27) ** 0.5)).requires_grad_(True)
    y64 = F.conv2d(x, w, None, padding=1)
    yq = q(y64.detach())
    y = y64 + (yq - y64).detach()                                                # value = the stored tensor, gradient = the convolution's
    gamma = detgen.normal((C,), seed + 3, mean=1.0, std=0.2).double(); beta = detgen.normal((C,), seed + 4, std=0.3).double()
    out = torch.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    g = q(detgen.normal((B, C, H, W), seed + 5, std=0.05))
    out.backward(g)
    mean = yq.mean((0, 2, 3)); invstd = (yq.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    scale = gamma * invstd; shift = beta - mean * scale
    zL = scale.view(1, C, 1, 1) * yq + shift.view(1, C, 1, 1)
    gz = g * (zL > 0)
    xhat = (yq - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
    n = B * H * W
    coef = torch.stack([scale, gz.sum((0, 2, 3)) / n, (gz * xhat).sum((0, 2, 3)) / n]).float()
    stats = torch.stack([scale, shift, mean, invstd]).float()
    # dy as the kernel forms it (csrc/wm_common.h wm_bn_fold; tests/test_gpu_bwd_oracle.py reference (B)), rounded to the storage type
    f = lambda t: t.float().double()           # noqa: E731
    v = lambda t: t.double().view(1, C, 1, 1)  # noqa: E731
    ca, c1, c2, mean32, invstd32 = coef[0], coef[1], coef[2], stats[2], stats[3]
    k2 = f(ca.double() * f(invstd32.double() * c2.double()))
    k3 = f(k2 * mean32.double() - f(ca.double() * c1.double()))
    dyq = q(f(v(ca) * gz + f(v(k3) - v(k2) * yq)).float())
    with torch.no_grad():
        dxB = torch.nn.grad.conv2d_input(x.shape, w.detach(), dyq, padding=1)
        dwB = torch.nn.grad.conv2d_weight(x.detach(), w.shape, dyq, padding=1)
    band = F.max_pool2d((zL.abs() < 1e-6).any(1, keepdim=True).double(), 3, 1, 1) > 0    # a ReLU decision at f32 round-off: its 3x3 neighbourhood of dx
    dy = scale.view(1, C, 1, 1) * (gz - coef[1].double().view(1, C, 1, 1) - xhat * coef[2].double().view(1, C, 1, 1))    # the exact graph's dy
    return dict(dy=dy.detach(), x=x.detach(), w=w.detach(), yq=yq, g=(gz if premasked else g), stats=stats, coef=coef, dx=x.grad.detach(), dw=w.grad.detach(),
                dxB=dxB, dwB=dwB, ok=~band)


@pytest.mark.parametrize("case", [(16, 256, 256, torch.bfloat16, True), (16, 256, 256, torch.float16, True), (16, 256, 256, torch.bfloat16, False),
                                  (2, 32, 48, torch.float16, False)])
def test_bwd_ws16_against_fp64_autograd(case):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, dt, premasked = case
    assert ops.conv3x3_bwd_fused16_supported((B, H, W, C), dt)
    o = _first_layer_operands(B, H, W, dt, 5300, premasked)
    u = _u(dt)
    wpt = ops.pack_w3x3(o["w"].float().cuda(), C, 16, dt, transpose=True)
    dw = torch.zeros(C, 3, 3, 3, device="cuda")
    dx = ops.conv3x3_bwd_fused16(_nhwc(o["g"], dt), _nhwc(o["yq"], dt), o["stats"].cuda().contiguous(), o["coef"].cuda().contiguous(), wpt,
                                 _nhwc(o["x"], dt, 16), dw, False, premasked=premasked)
    torch.cuda.synchronize()
    assert dx.shape == (B, H, W, 16)
    assert dx[..., 3:].abs().max().item() == 0.0                                 # the 13 padding channels of the image tensor get no gradient
    got = _nchw64(dx[..., :3])
    ok = o["ok"]
    assert ok.double().mean().item() > 0.99
    # ---- dx against the exact graph (A): the dy rounding noise through a 576-term GEMM + the storage rounding
    ref = o["dx"]
    err = (got - ref).abs() * ok
    rel_l2 = (err.pow(2).sum() / ref.pow(2).sum()).sqrt().item()
    assert rel_l2 < u and err.max().item() < 2 * u * ref.abs().max().item(), (rel_l2, err.max().item())
    # (B) against the fp64 GEMM of the dy the kernel rounds: the correctly rounded value up to f32 accumulation
    dxB = o["dxB"]
    errB = (got - dxB).abs() * ok
    tight = 1.02 * u * dxB.abs() + 1e-5 * dxB.abs().max()
    relB = (errB.pow(2).sum() / dxB.pow(2).sum()).sqrt().item()
    frac = (errB > tight).double().mean().item()
    print(f"[bwd_ws16 vs fp64 {dt} premasked={premasked} {B}x{H}x{W}] dx: (A) relative L2 {rel_l2:.3e}; (B) {relB:.3e} = {relB / u:.3f} u, beyond the tight bound: {frac:.2e}")
    assert relB < 0.5 * u
    # (f16: one f16 ulp of dy in ~1 element of 50,000, as in bwd_ws.hip -- DESIGN section 7 -- moves its 3x3 neighbourhood of dx)
    assert frac <= (0.0 if dt == torch.bfloat16 else 1e-3) and (errB <= tight + 4e-4 * dxB.abs().max()).all()
    # ---- dW: (B) only f32 accumulation left; (A) within the dy rounding noise
    dB = (dw.double().cpu() - o["dwB"]).abs()
    relWB = (dB.pow(2).sum() / o["dwB"].pow(2).sum()).sqrt().item()
    dA = (dw.double().cpu() - o["dw"]).abs()
    relWA = (dA.pow(2).sum() / o["dw"].pow(2).sum()).sqrt().item()
    print(f"    dW: (B) relative L2 {relWB:.3e}, max {dB.max().item() / o['dwB'].abs().max().item():.3e} of max|dW|; (A) relative L2 {relWA:.3e}")
    assert relWB < 1e-4 and dB.max().item() < 2e-4 * o["dwB"].abs().max().item()
    # (A): dW is a sum over ~1e6 pixels of dy x image that largely cancels on zero-mean random data, while the rounding of dy to the
    # storage type does not (measured relative L2 against (A): 4.5e-2 in bf16, 6e-4 in f16 at full size) -- so (A) is held to the NOISE
    # MODEL of that rounding, element by element, as tests/test_gpu_bwd_oracle.py does: sigma = R u sqrt(sum dy^2 x^2) plus the
    # mantissa-grid part 0.15 u sum |dy x|
    R = 0.43
    with torch.no_grad():
        sig = R * u * torch.nn.grad.conv2d_weight(o["x"].pow(2), o["dw"].shape, o["dy"].pow(2), padding=1).sqrt()
        l1w = torch.nn.grad.conv2d_weight(o["x"].abs(), o["dw"].shape, o["dy"].abs(), padding=1)
    print(f"    dW (A): max |diff| / sigma {(dA / sig).max().item():.2f}, max {(dA / l1w).max().item():.2e} of sum|dy x|")
    assert (dA <= 6 * sig + 0.15 * u * l1w).all()
