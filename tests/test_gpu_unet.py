"""GPU parity of the UNet tamper-localisation head (HIP path, f32 compute) against the golden
vectors generated from the reference's network/UNet.py and against the oracle; plus unit checks of
the pooling / up-convolution kernels against plain PyTorch ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detgen
from oracle import unet_ref

pytestmark = pytest.mark.gpu


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_fwd_bwd(dtype):
    from video_watermarking_forgery_detection_amd import ops
    B, C, H, W = 2, 64, 12, 20
    y = detgen.normal((B, C, H, W), 1)
    if dtype == torch.bfloat16:
        y = y.bfloat16().float()
    sc = detgen.normal((C,), 2, std=0.3, mean=1.0); sh = detgen.normal((C,), 3, std=0.3)
    a = torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dtype).float().requires_grad_(True)
    pooled_ref = F.max_pool2d(a, 2, 2)
    gp = detgen.normal((B, C, H // 2, W // 2), 4); gs = detgen.normal((B, C, H, W), 5)
    if dtype == torch.bfloat16:
        gp = gp.bfloat16().float(); gs = gs.bfloat16().float()
    pooled_ref.backward(gp)
    cat = torch.zeros(B, H, W, 2 * C, device="cuda", dtype=dtype)
    pooled = ops.bnrelu_maxpool2(nhwc(y, dtype), sc.cuda(), sh.cuda(), C, act_out=cat, act_c0=C)
    t0 = 1e-6 if dtype == torch.float32 else 1e-2  # bf16: fma vs mul+add may round the activation one ulp apart
    torch.testing.assert_close(pooled.float().cpu().permute(0, 3, 1, 2), pooled_ref.detach(), rtol=t0, atol=t0)
    torch.testing.assert_close(cat[..., C:].float().cpu().permute(0, 3, 1, 2), a.detach(), rtol=t0, atol=t0)
    gcat = torch.zeros(B, H, W, 2 * C, dtype=dtype); gcat[..., C:] = gs.permute(0, 2, 3, 1).to(dtype)
    g = ops.maxpool2_bwd(nhwc(y, dtype), sc.cuda(), sh.cuda(), nhwc(gp, dtype), gcat.cuda(), C, C)
    ref = (a.grad + gs)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(g.float().cpu().permute(0, 3, 1, 2), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 6, 10, 64, 32), (1, 4, 4, 512, 256), (2, 24, 20, 128, 64), (3, 17, 13, 64, 64), (2, 10, 12, 64, 16)])
def test_upconv_fwd_bwd(dtype, case):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, Cout = case
    x = detgen.normal((B, Cin, H, W), 10)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    sc = detgen.normal((Cin,), 11, std=0.3, mean=1.0); sh = detgen.normal((Cin,), 12, std=0.3)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    up = torch.nn.ConvTranspose2d(Cin, Cout, 2, 2)
    detgen.fill_module(up)
    ref = up(a)
    gy = detgen.normal((B, Cout, 2 * H, 2 * W), 13)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    ref.backward(gy)
    cat = torch.zeros(B, 2 * H, 2 * W, 2 * Cout, device="cuda", dtype=dtype)
    ops.upconv2x2_fwd(nhwc(x, dtype), sc.cuda(), sh.cuda(), up.weight.detach().cuda(), up.bias.detach().cuda(), cat, 0)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(cat[..., :Cout].float().cpu().permute(0, 3, 1, 2), ref.detach(), rtol=tol, atol=tol)
    assert cat[..., Cout:].abs().max().item() == 0
    gcat = torch.zeros(B, 2 * H, 2 * W, 2 * Cout, dtype=dtype); gcat[..., :Cout] = gy.permute(0, 2, 3, 1).to(dtype)
    dw = torch.zeros_like(up.weight).cuda(); db = torch.zeros_like(up.bias).cuda()
    gx = ops.upconv2x2_bwd(nhwc(x, dtype), sc.cuda(), sh.cuda(), up.weight.detach().cuda(), gcat.cuda(), 0, dw, db, False)
    torch.testing.assert_close(gx.float().cpu().permute(0, 3, 1, 2), a.grad, rtol=tol, atol=tol)
    t2 = 1e-4 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(dw.cpu(), up.weight.grad, rtol=t2, atol=t2 * up.weight.grad.abs().max().item())
    torch.testing.assert_close(db.cpu(), up.bias.grad, rtol=t2, atol=t2 * up.bias.grad.abs().max().item())
    # accumulate=True adds onto what is there
    dw2 = dw.clone(); db2 = db.clone()
    ops.upconv2x2_bwd(nhwc(x, dtype), sc.cuda(), sh.cuda(), up.weight.detach().cuda(), gcat.cuda(), 0, dw2, db2, True)
    torch.testing.assert_close(dw2, 2 * dw, rtol=1e-5, atol=1e-5 * dw.abs().max().item())
    torch.testing.assert_close(db2, 2 * db, rtol=1e-5, atol=1e-5 * db.abs().max().item())


def test_unet_golden(golden):
    import video_watermarking_forgery_detection_amd as wm
    from video_watermarking_forgery_detection_amd.network import UNet
    g = golden("unet")
    net = detgen.fill_module(UNet(3, 1, 32)).cuda().train()
    wm.set_compute_dtype(net, torch.float32)
    assert int(g["param_count"]) == sum(p.numel() for p in net.parameters()) == 7763041
    for (B, H) in ((1, 32), (2, 64)):
        net.zero_grad()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        key = f"unet_{B}x{H}"
        x = detgen.uniform((B, 3, H, H), 3000 + H).cuda().requires_grad_(True)
        y = net(x)
        (y * detgen.normal((B, 1, H, H), 8000 + H).cuda()).sum().backward()
        assert rel(y, g[key + "/y"]) < 1e-3, key                      # integer tamper masks follow: threshold below
        assert ((y.detach().cpu().numpy() > 0.5) != (g[key + "/y"] > 0.5)).mean() < 1e-3
        assert rel(x.grad, g[key + "/gx"]) < 5e-2, key                 # fp32 conditioning of a 18-conv BN chain
        bad = 0
        for n, p in net.named_parameters():
            ref = float(g[f"{key}/gnorm/{n}"])
            got = p.grad.norm().item()
            if abs(got - ref) > 5e-2 * ref + 1e-6:
                bad += 1
        assert bad <= 2, bad


def test_unet_vs_oracle_intermediate_size():
    import video_watermarking_forgery_detection_amd as wm
    from video_watermarking_forgery_detection_amd.network import UNet
    net = detgen.fill_module(UNet(3, 1, 32)).cuda().train()
    wm.set_compute_dtype(net, torch.float32)
    ref = detgen.fill_module(unet_ref.UNet(3, 1, 32)).train()
    x = detgen.uniform((1, 3, 48, 80), 9)
    gy = detgen.normal((1, 1, 48, 80), 10)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr); (yr * gy).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = net(xg); (y * gy.cuda()).sum().backward()
    assert rel(y, yr) < 1e-3
    num, den = 0.0, 0.0
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        num += (p.grad.cpu() - q.grad).norm().item() ** 2; den += q.grad.norm().item() ** 2
    assert (num / den) ** 0.5 < 3e-2
    # running statistics / counters follow torch
    for (n, b), (_, c) in zip(net.named_buffers(), ref.named_buffers()):
        if n.endswith("num_batches_tracked"):
            assert int(b) == int(c) == 1
        else:
            assert rel(b, c) < 1e-3, n


def test_unet_bf16_runs_and_bad_shape():
    import video_watermarking_forgery_detection_amd as wm
    from video_watermarking_forgery_detection_amd.network import UNet
    net = detgen.fill_module(UNet(3, 1, 32)).cuda().train()
    ref = detgen.fill_module(unet_ref.UNet(3, 1, 32)).train()
    x = detgen.uniform((2, 3, 64, 64), 11)
    y = net(x.cuda())
    assert rel(y, ref(x)) < 1e-1  # bf16 activations through 18 convolutions; the parity gate is the f32 path above
    with pytest.raises(RuntimeError, match="divisible by 16"):
        net(torch.zeros(1, 3, 40, 40, device="cuda"))
