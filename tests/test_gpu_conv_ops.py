"""GPU parity of the conv-path HIP kernels (through the C ABI) against plain PyTorch fp32 CPU
ops of the same definition: conv3x3 implicit GEMM (fwd, dgrad, wgrad), BatchNorm statistics /
backward, heads, Adam.  f32 path: tight tolerance (exact-f32 MFMA, different summation order);
bf16 path: tolerance of bf16 inputs (2^-8 relative) documented per assert."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detgen

pytestmark = pytest.mark.gpu


def nhwc(x, dtype, ld=None):
    """[B,C,H,W] f32 cpu -> [B,H,W,ld] cuda (zero padded channels)."""
    B, C, H, W = x.shape
    ld = ld or C
    out = torch.zeros(B, H, W, ld, dtype=dtype, device="cuda")
    out[..., :C] = x.permute(0, 2, 3, 1).to(dtype).cuda()
    return out


def nchw(t, C):
    return t[..., :C].float().permute(0, 3, 1, 2).cpu()


CASES = [  # B, H, W, Cin, Cout
    (2, 20, 37, 16, 64),
    (1, 32, 32, 64, 64),
    (2, 16, 16, 112, 64),
    (1, 33, 16, 64, 32),
    (2, 50, 70, 64, 32),      # persistent kernel, 32 output channels (image dgrads, the 30-channel layer)
    (1, 16, 48, 32, 128),
    (2, 40, 36, 128, 128),    # streamed-filter kernel (Cin > 64), ragged tiles, two output tiles
    (1, 20, 21, 256, 64),
    (3, 100, 90, 64, 64),     # persistent 64-channel kernel, ragged tiles
    (2, 256, 256, 64, 64),    # persistent kernel, 2 tiles per workgroup
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv3x3_fwd_and_stats(dtype, case):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, Cout = case
    x = detgen.normal((B, Cin, H, W), 1)
    w = detgen.normal((Cout, Cin, 3, 3), 2, std=(2.0 / (9 * Cin)) ** 0.5)
    bias = detgen.normal((Cout,), 3, std=0.1)
    if dtype == torch.bfloat16:  # compare against the same bf16-rounded operands
        x = x.bfloat16().float(); w = w.bfloat16().float()
    ref = F.conv2d(x, w, bias, padding=1)
    wp = ops.pack_w3x3(w.cuda(), Cout, Cin, dtype)
    y, st = ops.conv3x3_fwd(nhwc(x, dtype), wp, bias.cuda(), None, None, True)
    got = nchw(y, Cout)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2  # bf16 output rounding: 2^-8 * |y| (|y| <~ 4)
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    # statistics come from the f32 accumulators: tight in both modes
    s = st.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(s[1], (ref * ref).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_fused_input_transform(dtype):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, Cout = 2, 24, 24, 64, 64
    x = detgen.normal((B, Cin, H, W), 4)
    w = detgen.normal((Cout, Cin, 3, 3), 5, std=0.06)
    sc = detgen.normal((Cin,), 6, std=0.5, mean=1.0)
    sh = detgen.normal((Cin,), 7, std=0.5)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float(); w = w.bfloat16().float()
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if dtype == torch.bfloat16:
        a = a.bfloat16().float()  # the kernel rounds the activated value when it stages it
    ref = F.conv2d(a, w, None, padding=1)  # zero padding AFTER the activation
    wp = ops.pack_w3x3(w.cuda(), Cout, Cin, dtype)
    y, _ = ops.conv3x3_fwd(nhwc(x, dtype), wp, None, sc.cuda(), sh.cuda(), False)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(nchw(y, Cout), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 20, 37, 64, 64), (1, 16, 16, 16, 64), (1, 24, 16, 64, 32), (5, 80, 112, 64, 64)])
def test_conv3x3_dgrad(dtype, case):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, Cout = case
    w = detgen.normal((Cout, Cin, 3, 3), 8, std=0.06)
    dy = detgen.normal((B, Cout, H, W), 9)
    if dtype == torch.bfloat16:
        w = w.bfloat16().float(); dy = dy.bfloat16().float()
    x = torch.zeros(B, Cin, H, W, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    CinP = max(32, Cin)  # dgrad output channels must be a multiple of 32
    CoutP = Cout if Cout % 32 == 0 else 32
    wpt = ops.pack_w3x3(w.cuda(), CoutP, CinP, dtype, transpose=True)     # [9][CinP][CoutP]
    gx, _ = ops.conv3x3_fwd(nhwc(dy, dtype, CoutP), wpt, None, None, None, False)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(nchw(gx, Cin), x.grad, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 20, 37, 64, 64), (1, 16, 16, 16, 64), (2, 32, 32, 64, 32), (1, 16, 32, 128, 64), (3, 48, 48, 64, 64)])
def test_conv3x3_wgrad(dtype, case):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, Cout = case
    x = detgen.normal((B, Cin, H, W), 10)
    dy = detgen.normal((B, Cout, H, W), 11)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float(); dy = dy.bfloat16().float()
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    dw = torch.full((Cout, Cin, 3, 3), 7.0, device="cuda")
    ops.conv3x3_wgrad(nhwc(x, dtype), Cin, None, None, nhwc(dy, dtype), dw, accumulate=False)
    scale = w.grad.abs().max().item()
    tol = 1e-5 if dtype == torch.float32 else 1e-3   # products of bf16-exact inputs accumulate in f32
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=tol, atol=tol * scale)
    # accumulate=True adds on top
    ops.conv3x3_wgrad(nhwc(x, dtype), Cin, None, None, nhwc(dy, dtype), dw, accumulate=True)
    torch.testing.assert_close(dw.cpu(), 2 * w.grad, rtol=tol, atol=2 * tol * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_with_transform_and_perm(dtype):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cout = 2, 16, 16, 64
    Cref, CinX = 10, 16
    perm = [9, 8, 7, 6, 5, 4, 3, 2, 1, 0]  # packed position of each reference channel
    xr = detgen.normal((B, Cref, H, W), 12)
    sc = detgen.normal((CinX,), 13, std=0.5, mean=1.0)
    sh = detgen.normal((CinX,), 14, std=0.3)
    dy = detgen.normal((B, Cout, H, W), 15)
    xp = torch.zeros(B, CinX, H, W)
    for ci, p in enumerate(perm):
        xp[:, p] = xr[:, ci]
    if dtype == torch.bfloat16:
        xp = xp.bfloat16().float(); dy = dy.bfloat16().float()
    a = torch.relu(xp * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if dtype == torch.bfloat16:
        a = a.bfloat16().float()
    a_ref = torch.stack([a[:, p] for p in perm], 1)
    w = torch.zeros(Cout, Cref, 3, 3, requires_grad=True)
    F.conv2d(a_ref, w, None, padding=1).backward(dy)
    dw = torch.zeros(Cout, Cref, 3, 3, device="cuda")
    ops.conv3x3_wgrad(nhwc(xp, dtype), CinX, sc.cuda(), sh.cuda(), nhwc(dy, dtype), dw, False,
                      perm_dev=torch.tensor(perm, dtype=torch.int32, device="cuda"))
    tol = 1e-5 if dtype == torch.float32 else 1e-3
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=tol, atol=tol * w.grad.abs().max().item())
    # forward pack with the same permutation
    wv = detgen.normal((Cout, Cref, 3, 3), 16, std=0.1)
    if dtype == torch.bfloat16:
        wv = wv.bfloat16().float()
    wp = ops.pack_w3x3(wv.cuda(), Cout, CinX, dtype, perm=perm)
    y, _ = ops.conv3x3_fwd(nhwc(xp, dtype), wp, None, sc.cuda(), sh.cuda(), False)
    ref = F.conv2d(a_ref, wv, None, padding=1)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(nchw(y, Cout), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,CP", [(64, 64), (30, 32)])
def test_bn_train_fwd_bwd(dtype, C, CP):
    """conv output y -> BN(train) -> ReLU, forward statistics/running stats and the full backward."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W = 3, 20, 24
    y = detgen.normal((B, C, H, W), 20, std=1.5, mean=0.3)
    g = detgen.normal((B, C, H, W), 21)
    if dtype == torch.bfloat16:
        y = y.bfloat16().float(); g = g.bfloat16().float()
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(detgen.normal((C,), 22, std=0.2, mean=1.0)); bn.bias.copy_(detgen.normal((C,), 23, std=0.2))
    yr = y.clone().requires_grad_(True)
    out = torch.relu(bn(yr))
    out.backward(g)
    # device side: statistics as the conv epilogue would emit them (one partial row)
    yd = nhwc(y, dtype, CP)
    part = torch.zeros(1, 2, CP, device="cuda")
    part[0, 0, :C] = y.sum((0, 2, 3)).cuda(); part[0, 1, :C] = (y * y).sum((0, 2, 3)).cuda()
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    gamma, beta = bn.weight.detach().cuda(), bn.bias.detach().cuda()
    st = ops.bn_finalize(part, C, CP, B * H * W, gamma, beta, rm, rv, 0.1, 1e-5)
    torch.testing.assert_close(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    act = torch.relu(yd[..., :C].float() * st[0, :C] + st[1, :C]).permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(act, out.detach(), rtol=1e-4, atol=1e-4)
    dgamma = torch.zeros(C, device="cuda"); dbeta = torch.zeros(C, device="cuda"); dbias = torch.zeros(C, device="cuda")
    dy = ops.bn_bwd(nhwc(g, dtype, CP), None, yd, st, C, gamma, dgamma, dbeta, False, dbias)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(nchw(dy, C), yr.grad, rtol=tol, atol=tol)
    torch.testing.assert_close(dgamma.cpu(), bn.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dbeta.cpu(), bn.bias.grad, rtol=1e-4, atol=1e-3)
    assert dbias.abs().max().item() < (1e-2 if dtype == torch.float32 else 1.0)  # sum(dy) == 0 up to rounding
    if CP > C:
        assert dy[..., C:].abs().max().item() == 0.0
    # gvec form: gradient of a global average pool
    gv = detgen.normal((B, C), 24)
    yr2 = y.clone().requires_grad_(True)
    torch.relu(bn(yr2)).mean((2, 3)).backward(gv)
    gvec = torch.zeros(B, CP, device="cuda"); gvec[:, :C] = (gv / (H * W)).cuda()
    dy2 = ops.bn_bwd(None, gvec, yd, st, C, gamma, dgamma, dbeta, False, None)
    torch.testing.assert_close(nchw(dy2, C), yr2.grad, rtol=tol, atol=tol * 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads(dtype):
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin = 2, 12, 20, 64
    y = detgen.normal((B, Cin, H, W), 30)
    if dtype == torch.bfloat16:
        y = y.bfloat16().float()
    sc = detgen.normal((Cin,), 31, std=0.3, mean=1.0); sh = detgen.normal((Cin,), 32, std=0.3)
    a = torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    # global average pool
    pooled = ops.bnrelu_avgpool(nhwc(y, dtype), sc.cuda(), sh.cuda())
    torch.testing.assert_close(pooled.cpu(), a.mean((2, 3)), rtol=1e-4, atol=1e-5)
    # 1x1 head, Cout = 3 and 1 (sigmoid)
    for Cout, act in ((3, 0), (1, 1)):
        w = detgen.normal((Cout, Cin), 33 + Cout, std=0.2).requires_grad_(True)
        b = detgen.normal((Cout,), 34 + Cout, std=0.2).requires_grad_(True)
        ar = a.clone().requires_grad_(True)
        lin = F.conv2d(ar, w.view(Cout, Cin, 1, 1), b)
        ref = torch.sigmoid(lin) if act else lin
        out = ops.conv1x1_head_fwd(nhwc(y, dtype), sc.cuda(), sh.cuda(), w.detach().cuda(), b.detach().cuda(), act)
        torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
        gout = detgen.normal((B, Cout, H, W), 40)
        lin.backward(gout)
        dw = torch.zeros(Cout, Cin, device="cuda"); db = torch.zeros(Cout, device="cuda")
        g = ops.conv1x1_head_bwd(nhwc(y, dtype), sc.cuda(), sh.cuda(), w.detach().cuda(), gout.cuda(), dw, db, False)
        tol = 1e-4 if dtype == torch.float32 else 1e-2
        torch.testing.assert_close(nchw(g, Cin), ar.grad, rtol=tol, atol=tol)
        torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-3)


def test_layout_roundtrip_and_concat():
    from video_watermarking_forgery_detection_amd import ops
    B, H, W = 2, 9, 13
    img = detgen.uniform((B, 3, H, W), 50)
    for dtype in (torch.float32, torch.bfloat16):
        x16 = torch.full((B, H, W, 16), 5.0, device="cuda", dtype=dtype)
        ops.nchw_to_nhwc(img.cuda(), x16, 0, 13)
        ref = torch.zeros(B, H, W, 16); ref[..., :3] = img.permute(0, 2, 3, 1)
        torch.testing.assert_close(x16.float().cpu(), ref.to(dtype).float())
        back = ops.nhwc_to_nchw(x16, 3, 0)
        torch.testing.assert_close(back.cpu(), img.to(dtype).float())
        cat = torch.full((B, H, W, 112), 9.0, device="cuda", dtype=dtype)
        msg = detgen.bits((B, 30), 51)
        feat = detgen.normal((B, 64, H, W), 52)
        sc = detgen.normal((64,), 53, mean=1.0, std=0.2); sh = detgen.normal((64,), 54, std=0.2)
        ops.bnrelu_copy(nhwc(feat, dtype), sc.cuda(), sh.cuda(), cat, 0, 64)
        ops.broadcast_to_nhwc(msg.cuda(), cat, 64)
        ops.nchw_to_nhwc(img.cuda(), cat, 94, 15)
        c = cat.float().cpu()
        fr = torch.relu(feat.to(dtype).float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
        torch.testing.assert_close(c[..., :64], fr.to(dtype).float(), rtol=1e-2, atol=1e-2)
        assert torch.equal(c[..., 64:94], msg.view(B, 1, 1, 30).expand(B, H, W, 30))
        torch.testing.assert_close(c[..., 94:97], img.permute(0, 2, 3, 1).to(dtype).float())
        assert c[..., 97:].abs().max() == 0
        # the one-pass form used by the encoder writes the very same row
        cat2 = torch.full((B, H, W, 112), 7.0, device="cuda", dtype=dtype)
        ops.concat_full(nhwc(feat, dtype), sc.cuda(), sh.cuda(), msg.cuda(), img.cuda(), cat2, 64)
        assert torch.equal(cat2, cat)
        # ... and so do the two halves it replaced (wm_bnrelu_copy + wm_concat_tail)
        cat3 = torch.full((B, H, W, 112), 3.0, device="cuda", dtype=dtype)
        ops.bnrelu_copy(nhwc(feat, dtype), sc.cuda(), sh.cuda(), cat3, 0, 64)
        ops.concat_tail(msg.cuda(), img.cuda(), cat3, 64)
        assert torch.equal(cat3, cat)


def test_adam_matches_torch():
    from video_watermarking_forgery_detection_amd import ops
    n = 10007
    p0 = detgen.normal((n,), 60);
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr])
    p = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = detgen.normal((n,), 60 + step)
        pr.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
        torch.testing.assert_close(p.cpu(), pr.detach(), rtol=1e-6, atol=1e-6)
    # AdamW
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=2e-4, betas=(0.9, 0.99), weight_decay=0.01)
    p = p0.clone().cuda(); m.zero_(); v.zero_()
    for step in range(1, 3):
        g = detgen.normal((n,), 70 + step)
        pr.grad = g.clone(); opt.step()
        ops.adam_step(p, g.cuda(), m, v, 2e-4, 0.9, 0.99, 1e-8, 0.01, step, decoupled=True)
        torch.testing.assert_close(p.cpu(), pr.detach(), rtol=1e-6, atol=1e-6)


def test_mse_and_sumsq():
    from video_watermarking_forgery_detection_amd import ops
    a = detgen.normal((3, 3, 40, 40), 80); b = detgen.normal((3, 3, 40, 40), 81)
    part, grad = ops.mse_fwd_bwd(a.cuda(), b.cuda(), 2.0 / a.numel())
    torch.testing.assert_close(part.sum().cpu() / a.numel(), F.mse_loss(a, b), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(grad.cpu(), 2 * (a - b) / a.numel())
    torch.testing.assert_close(ops.sumsq(a.cuda()).sum().cpu(), (a * a).sum(), rtol=1e-5, atol=1e-5)


def test_shape_errors():
    from video_watermarking_forgery_detection_amd import ops
    x = torch.zeros(1, 8, 8, 24, device="cuda", dtype=torch.bfloat16)  # Cin=24 not a multiple of 16
    wp = torch.zeros(9, 64, 24, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple"):
        ops.conv3x3_fwd(x, wp, None, None, None, False)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.conv3x3_fwd(x.cpu(), wp, None, None, None, False)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("case", [(2, 40, 52, 64, 64), (1, 20, 37, 32, 64), (2, 33, 18, 16, 64)])
def test_ws_conv_mfma_shapes_agree(case, variant, debug_lib):
    """the wave-specialised bf16 kernel: 16x16x32 consumers (default) and 32x32x16 consumers (debug knob) against
    conv2d on the same bf16-rounded operands, with the fused BN+ReLU input transform and the statistics."""
    import ctypes
    from video_watermarking_forgery_detection_amd import ops, _lib
    B, H, W, Cin, Cout = case
    x = detgen.normal((B, Cin, H, W), 61).bfloat16().float()
    w = detgen.normal((Cout, Cin, 3, 3), 62, std=(2.0 / (9 * Cin)) ** 0.5).bfloat16().float()
    bias = detgen.normal((Cout,), 63, std=0.1)
    sc = detgen.normal((Cin,), 64, mean=1.0, std=0.2); sh = detgen.normal((Cin,), 65, std=0.3)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).bfloat16().float()
    ref = F.conv2d(a, w, bias, padding=1)
    wp = ops.pack_w3x3(w.cuda(), Cout, Cin, torch.bfloat16)
    L = debug_lib   # the -DWM_DEBUG build: the release library has no A/B switches
    L.wm_debug_ws_variant(ctypes.c_int(variant))
    try:
        y, st = ops.conv3x3_fwd(nhwc(x, torch.bfloat16), wp, bias.cuda(), sc.cuda(), sh.cuda(), True)
    finally:
        L.wm_debug_ws_variant(ctypes.c_int(0))
    torch.testing.assert_close(nchw(y, Cout), ref, rtol=1.5e-2, atol=1.5e-2)
    s = st.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(s[1], (ref * ref).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 32, 48), (1, 64, 16)])
def test_ws_conv_whole_tile_form_is_the_masked_form(case, dt, debug_lib):
    """shapes made of whole 16 x 16 tiles take the forward kernel's form without the inside-the-image mask (conv3x3_ws.hip, WHOLE): the
    output and the BatchNorm partial sums equal the masked form's (debug knob 11) bit for bit, and conv2d's within the 16-bit bound."""
    import ctypes
    from video_watermarking_forgery_detection_amd import ops
    B, H, W = case
    C = 64
    x = detgen.normal((B, C, H, W), 71).to(dt).float()
    w = detgen.normal((C, C, 3, 3), 72, std=(2.0 / (9 * C)) ** 0.5).to(dt).float()
    bias = detgen.normal((C,), 73, std=0.1)
    sc = detgen.normal((C,), 74, mean=1.0, std=0.2); sh = detgen.normal((C,), 75, std=0.3)
    wp = ops.pack_w3x3(w.cuda(), C, C, dt)
    xin = nhwc(x, dt)
    L = debug_lib
    y0, st0 = ops.conv3x3_fwd(xin, wp, bias.cuda(), sc.cuda(), sh.cuda(), True)
    L.wm_debug_ws_variant(ctypes.c_int(11))
    try:
        y1, st1 = ops.conv3x3_fwd(xin, wp, bias.cuda(), sc.cuda(), sh.cuda(), True)
    finally:
        L.wm_debug_ws_variant(ctypes.c_int(0))
    assert torch.equal(y0, y1) and torch.equal(st0, st1)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float()
    ref = F.conv2d(a, w, bias, padding=1)
    tol = 1.5e-2 if dt == torch.bfloat16 else 2e-3
    torch.testing.assert_close(nchw(y0, C), ref, rtol=tol, atol=tol)
    torch.testing.assert_close(st0.sum(0).cpu()[0], ref.sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


def test_ws_first_layer_whole_tile_form_is_the_masked_form(debug_lib):
    """the same for the image-fed first layers' forward (16 -> 64 channels, no input transform, statistics)"""
    import ctypes
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, C = 2, 48, 32, 16, 64
    dt = torch.bfloat16
    x = detgen.normal((B, Cin, H, W), 81).to(dt).float()
    w = detgen.normal((C, Cin, 3, 3), 82, std=(2.0 / (9 * Cin)) ** 0.5).to(dt).float()
    bias = detgen.normal((C,), 83, std=0.1)
    wp = ops.pack_w3x3(w.cuda(), C, Cin, dt)
    xin = nhwc(x, dt)
    L = debug_lib
    y0, st0 = ops.conv3x3_fwd(xin, wp, bias.cuda(), None, None, True)
    L.wm_debug_ws_variant(ctypes.c_int(11))
    try:
        y1, st1 = ops.conv3x3_fwd(xin, wp, bias.cuda(), None, None, True)
    finally:
        L.wm_debug_ws_variant(ctypes.c_int(0))
    assert torch.equal(y0, y1) and torch.equal(st0, st1)
    ref = F.conv2d(x, w, bias, padding=1)
    torch.testing.assert_close(nchw(y0, C), ref, rtol=1.5e-2, atol=1.5e-2)
    torch.testing.assert_close(st0.sum(0).cpu()[1], (ref * ref).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


def test_pack_plan_matches_single_packs():
    """ops.PackPlan (one wm_pack_w3x3_batch launch for a whole network) == wm_pack_w3x3 per conv, including the
    permuted / transposed (dgrad) packs, and follows parameter updates after refresh()."""
    from video_watermarking_forgery_detection_amd import ops
    ws = [detgen.normal((64, 16, 3, 3), 71).cuda(), detgen.normal((64, 64, 3, 3), 72).cuda(), detgen.normal((30, 64, 3, 3), 73).cuda(),
          detgen.normal((64, 97, 3, 3), 74).cuda()]
    perm = list(range(64, 94)) + list(range(0, 64)) + list(range(94, 97))
    reqs = [(ws[0], 64, 16, None, False), (ws[1], 64, 64, None, False), (ws[1], 64, 64, None, True), (ws[2], 32, 64, None, False),
            (ws[2], 32, 64, None, True), (ws[3], 64, 112, perm, False), (ws[3], 64, 64, perm, True)]
    plan = ops.PackPlan()
    for w, a, b, p, t in reqs:       # registration pass: packs one by one
        plan.get(w, a, b, torch.bfloat16, perm=p, transpose=t)
    for rnd in range(2):
        plan.refresh()
        for w, a, b, p, t in reqs:
            got = plan.get(w, a, b, torch.bfloat16, perm=p, transpose=t)
            ref = ops.pack_w3x3(w, a, b, torch.bfloat16, perm=p, transpose=t)
            assert torch.equal(got, ref)
        plan.invalidate()
        for w in ws:                 # "optimiser step"
            w.mul_(0.5).add_(0.01)


def test_fused_loss_kernels_match_torch():
    """wm_bce_logits / wm_message_loss against the torch definitions the reference uses (hidden.py:68-111)."""
    from video_watermarking_forgery_detection_amd import ops
    for n, seed in ((16, 91), (1, 92), (700, 93)):
        x = (detgen.normal((n, 1), seed, std=3.0)).cuda()
        for target in (0.0, 1.0):
            xr = x.clone().requires_grad_(True)
            ref = F.binary_cross_entropy_with_logits(xr, torch.full_like(xr, target))
            ref.backward()
            loss, grad = ops.bce_logits(x, target, gscale=0.37)
            torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(grad.view_as(x), 0.37 * xr.grad, rtol=1e-5, atol=1e-7)
    d = (detgen.uniform((16, 30), 94) * 1.6 - 0.3).cuda()
    d[0, :4] = torch.tensor([0.5, 1.5, -0.5, 2.5], device="cuda")      # round-half-to-even cases
    m = detgen.bits((16, 30), 95).cuda()
    out, grad = ops.message_loss(d, m, 0.01)
    torch.testing.assert_close(out[0], ((d - m) ** 2).mean(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(out[1], (d.round().clamp(0, 1) - m).abs().sum() / d.numel(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(grad, (d - m) * 0.01, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("shape", [(16, 30, 30, 32), (16, 64, 1, 64), (3, 7, 5, 16)])
def test_linear_head_kernels_match_torch(shape):
    """wm_linear_head_fwd/bwd == nn.Linear on the pooled features and its autograd (decoder.py:26-34, discriminator.py:18-26)."""
    from video_watermarking_forgery_detection_amd import ops
    B, I, O, CP = shape
    pooled = detgen.normal((B, CP), 101).cuda()
    w = detgen.normal((O, I), 102, std=0.3).cuda(); bias = detgen.normal((O,), 103, std=0.1).cuda()
    g = detgen.normal((B, O), 104).cuda()
    out = ops.linear_head_fwd(pooled, w, bias, I)
    torch.testing.assert_close(out, pooled[:, :I] @ w.t() + bias, rtol=1e-5, atol=1e-5)
    for accumulate in (False, True):
        dw = torch.full((O, I), 0.5, device="cuda"); db = torch.full((O,), -0.25, device="cuda")
        gvec = ops.linear_head_bwd(pooled, w, g, dw, db, accumulate, CP, 1.0 / 64.0)
        base_w, base_b = (0.5, -0.25) if accumulate else (0.0, 0.0)
        torch.testing.assert_close(dw, base_w + g.t() @ pooled[:, :I], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(db, base_b + g.sum(0), rtol=1e-5, atol=1e-5)
        ref = torch.zeros(B, CP, device="cuda"); ref[:, :I] = (g @ w) / 64.0
        torch.testing.assert_close(gvec, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", [(2, 40, 36), (1, 17, 50), (3, 64, 64)])
def test_wgrad_with_fused_bn_backward_apply(case):
    """wm_conv3x3_wgrad_bnfused (dy formed inside the weight-gradient kernel from g, y and the BatchNorm constants) is
    bit-identical to wm_bn_bwd_apply followed by wm_conv3x3_wgrad: same arithmetic, same bf16 rounding of dy."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W = case
    C = 64
    x = nhwc(detgen.uniform((B, 3, H, W), 111), torch.bfloat16, 16)
    y = nhwc(detgen.normal((B, C, H, W), 112), torch.bfloat16)
    g = nhwc(detgen.normal((B, C, H, W), 113), torch.bfloat16)
    gamma = detgen.normal((C,), 114, mean=1.0, std=0.3).cuda(); beta = detgen.normal((C,), 115, std=0.3).cuda()
    yf = y.float()
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    scale = gamma * invstd
    stats = torch.stack([scale, beta - mean * scale, mean, invstd]).contiguous()
    dg0 = torch.zeros(C, device="cuda"); db0 = torch.zeros(C, device="cuda")
    dy = ops.bn_bwd(g, None, y, stats, C, gamma, dg0, db0, False, None)
    dw0 = torch.zeros(C, 3, 3, 3, device="cuda")
    ops.conv3x3_wgrad(x, 16, None, None, dy, dw0, False)
    dg1 = torch.zeros(C, device="cuda"); db1 = torch.zeros(C, device="cuda")
    coef = ops.bn_bwd_coef(g, None, y, stats, C, gamma, dg1, db1, False)
    dw1 = torch.full((C, 3, 3, 3), 0.25, device="cuda")
    ops.conv3x3_wgrad_bnfused(x, g, y, stats, coef, dw1, False)
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dw0, dw1)
    ops.conv3x3_wgrad_bnfused(x, g, y, stats, coef, dw1, True)      # accumulate
    torch.testing.assert_close(dw1, 2 * dw0, rtol=1e-6, atol=1e-6 * dw0.abs().max().item())


@pytest.mark.parametrize("case", [(2, 40, 36, 64), (1, 17, 50, 64), (3, 64, 64, 64), (2, 48, 33, 30), (17, 16, 16, 64)])
def test_pooled_layer_backward_with_fused_bn_apply(case):
    """wm_conv3x3_wgrad_gvfused / wm_conv3x3_dgrad_gvfused (a globally pooled ConvBNRelu: dy formed from gvec, y and the
    BatchNorm constants inside both consumer kernels) are bit-identical to wm_bn_bwd_apply followed by wm_conv3x3_wgrad /
    wm_conv3x3_fwd with the transposed filter.  Cout 64 (discriminator.py:14-22) and 30 -> 32 (decoder.py:16-24); the
    last case has tile runs that cross samples."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, C = case
    CP = 32 * ((C + 31) // 32)
    Cin = 64
    x = nhwc(detgen.normal((B, Cin, H, W), 121), torch.bfloat16)
    xs = detgen.normal((Cin,), 122, mean=1.0, std=0.3).cuda(); xt = detgen.normal((Cin,), 123, std=0.3).cuda()
    y = nhwc(detgen.normal((B, C, H, W), 124), torch.bfloat16, CP)
    gvec = torch.zeros(B, CP, device="cuda"); gvec[:, :C] = detgen.normal((B, C), 125).cuda() / (H * W)
    gamma = detgen.normal((C,), 126, mean=1.0, std=0.3).cuda(); beta = detgen.normal((C,), 127, std=0.3).cuda()
    w = detgen.normal((C, Cin, 3, 3), 128, std=0.05).cuda()
    yf = y.float()[..., :C]
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    scale = gamma * invstd
    stats = torch.zeros(4, CP, device="cuda")
    stats[0, :C] = scale; stats[1, :C] = beta - mean * scale; stats[2, :C] = mean; stats[3, :C] = invstd
    assert ops.conv3x3_gvfused_supported(Cin, CP, torch.bfloat16)
    dg0 = torch.zeros(C, device="cuda"); db0 = torch.zeros(C, device="cuda")
    dy = ops.bn_bwd(None, gvec, y, stats, C, gamma, dg0, db0, False, None)
    dw0 = torch.zeros(C, Cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad(x, Cin, xs, xt, dy, dw0, False)
    wpt = ops.pack_w3x3(w, CP, Cin, torch.bfloat16, transpose=True)
    dx0, _ = ops.conv3x3_fwd(dy, wpt, None, None, None, want_stats=False)
    dg1 = torch.zeros(C, device="cuda"); db1 = torch.zeros(C, device="cuda")
    coef = ops.bn_bwd_coef(None, gvec, y, stats, C, gamma, dg1, db1, False)
    dw1 = torch.full((C, Cin, 3, 3), 0.25, device="cuda")
    ops.conv3x3_wgrad_gvfused(x, xs, xt, gvec, y, stats, coef, dw1, False)
    dx1 = ops.conv3x3_dgrad_gvfused(y, wpt, gvec, stats, coef)
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert dw0.abs().max().item() > 0 and dx0.float().abs().max().item() > 0
    assert torch.equal(dw0, dw1)
    assert torch.equal(dx0, dx1)
    ops.conv3x3_wgrad_gvfused(x, xs, xt, gvec, y, stats, coef, dw1, True)      # accumulate
    torch.testing.assert_close(dw1, 2 * dw0, rtol=1e-6, atol=1e-6 * dw0.abs().max().item())


def test_pooled_layer_fusion_keeps_the_training_step(debug_lib):
    """the HiDDeN step with the fused pooled-layer backward gives the same losses and parameters as with the separate apply pass"""
    import ctypes
    from video_watermarking_forgery_detection_amd import _lib
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    L = debug_lib   # the -DWM_DEBUG build: the release library has no A/B switches
    outs = []
    try:
        L.wm_debug_bwd_fuse(ctypes.c_int(0))   # (the one-kernel backward sums over other tiles: test_one_kernel_backward_keeps_the_training_step)
        for on in (1, 0):
            L.wm_debug_gv_fuse(ctypes.c_int(on))
            torch.manual_seed(10)
            h = Hidden(HiDDenConfiguration(H=64, W=64), torch.device("cuda"), NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
            images = detgen.uniform((4, 3, 64, 64), 131).cuda(); messages = (detgen.uniform((4, 30), 132) > 0.5).float().cuda()
            for _ in range(2):
                losses, _ = h.train_on_batch([images, messages])
            outs.append((losses, [p.detach().clone() for p in list(h.encoder_decoder.parameters()) + list(h.discriminator.parameters())]))
    finally:
        L.wm_debug_gv_fuse(ctypes.c_int(1))
        L.wm_debug_bwd_fuse(ctypes.c_int(1))
    for k in outs[0][0]:
        assert outs[0][0][k] == outs[1][0][k], k
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(2, 40, 36, 64, False), (1, 17, 50, 64, True), (3, 64, 64, 30, True), (17, 16, 16, 64, False)])
def test_dgrad_that_reduces_the_feeding_layers_bn_backward_sums(case):
    """wm_conv3x3_dgrad_bwdstats: dx bit-identical to the plain dgrad (and to the apply-fused one), and the partial rows it
    emits finish (wm_bn_bwd_finalize_raw) into the same dgamma / dbeta / coef as wm_bn_bwd_reduce over (dx, ry)."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, C, pooled = case
    CP = 32 * ((C + 31) // 32)
    dy = nhwc(detgen.normal((B, C, H, W), 141), torch.bfloat16, CP)
    w = detgen.normal((C, 64, 3, 3), 142, std=0.05).cuda()
    wpt = ops.pack_w3x3(w, CP, 64, torch.bfloat16, transpose=True)
    ry = nhwc(detgen.normal((B, 64, H, W), 143, mean=0.4), torch.bfloat16)
    gamma = detgen.normal((64,), 144, mean=1.0, std=0.3).cuda(); beta = detgen.normal((64,), 145, std=0.3).cuda()
    yf = ry.float()
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    scale = gamma * invstd
    rstats = torch.stack([scale, beta - mean * scale, mean, invstd]).contiguous()
    assert ops.conv3x3_dgrad_bwdstats_supported(CP, 64, torch.bfloat16)
    if pooled:   # this layer is the pooled one: src = its raw output, the apply pass fused as well
        gvec = torch.zeros(B, CP, device="cuda"); gvec[:, :C] = detgen.normal((B, C), 146).cuda() / (H * W)
        g2 = detgen.normal((C,), 147, mean=1.0, std=0.3).cuda()
        st = torch.zeros(4, CP, device="cuda")
        y2 = dy.float()[..., :C]
        m2 = y2.mean((0, 1, 2)); i2 = torch.rsqrt(y2.var((0, 1, 2), unbiased=False) + 1e-5)
        st[0, :C] = g2 * i2; st[1, :C] = -m2 * g2 * i2; st[2, :C] = m2; st[3, :C] = i2
        coef = ops.bn_bwd_coef(None, gvec, dy, st, C, g2, torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), False)
        dx0 = ops.conv3x3_dgrad_gvfused(dy, wpt, gvec, st, coef)
        dx1, part = ops.conv3x3_dgrad_bwdstats(dy, wpt, ry, rstats[0], rstats[1], gvec, st, coef)
    else:
        dx0, _ = ops.conv3x3_fwd(dy, wpt, None, None, None, want_stats=False)
        dx1, part = ops.conv3x3_dgrad_bwdstats(dy, wpt, ry, rstats[0], rstats[1])
    assert dx0.float().abs().max().item() > 0
    # the kernel that reduces the feeding layer's sums also WRITES its gradient multiplied by that layer's ReLU mask (round 3: every consumer
    # applies the mask anyway, and the one-pass backward's premasked staging then needs no mask arithmetic)
    zf = rstats[0] * ry.float() + rstats[1]
    sure = zf.abs() > 1e-4                      # (the kernel evaluates z with one fma: the sign may differ within round-off of zero)
    assert torch.equal(torch.where(zf > 0, dx0, torch.zeros_like(dx0))[sure], dx1[sure]) and getattr(dx1, "_wm_masked", False)
    dg0 = torch.zeros(64, device="cuda"); db0 = torch.zeros(64, device="cuda")
    c0 = ops.bn_bwd_coef(dx0, None, ry, rstats, 64, gamma, dg0, db0, False)
    dg1 = torch.full((64,), 0.5, device="cuda"); db1 = torch.full((64,), -0.5, device="cuda")
    c1 = ops.bn_bwd_coef_raw(part, ry, rstats, 64, gamma, dg1, db1, False)
    # same addends; f32 partial sums in another order, and sum(gz*xhat) as invstd*(sum(gz*y) - mean*sum(gz))
    sc = dg0.abs().max().item() + db0.abs().max().item()
    torch.testing.assert_close(db1, db0, rtol=1e-4, atol=1e-5 * sc)
    torch.testing.assert_close(dg1, dg0, rtol=1e-4, atol=1e-5 * sc)
    torch.testing.assert_close(c1, c0, rtol=1e-4, atol=1e-5 * c0.abs().max().item())
    # accumulate (the finalisation treats the partial rows as scratch: emit them again)
    part = ops.conv3x3_dgrad_bwdstats(dy, wpt, ry, rstats[0], rstats[1], *((gvec, st, coef) if pooled else ()))[1]
    ops.bn_bwd_coef_raw(part, ry, rstats, 64, gamma, dg1, db1, True)
    torch.testing.assert_close(dg1, 2 * dg0, rtol=1e-4, atol=2e-5 * sc)


def test_fused_bn_backward_reduce_keeps_the_training_step(debug_lib):
    """the HiDDeN step with the BatchNorm-backward reduce passes folded into the dgrad epilogues: same losses, and parameters
    within f32 summation-order noise of the step with separate reduce passes"""
    import ctypes
    from video_watermarking_forgery_detection_amd import _lib
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    L = debug_lib   # the -DWM_DEBUG build: the release library has no A/B switches
    outs = []
    try:
        for on in (1, 0):
            L.wm_debug_bwdst(ctypes.c_int(on))
            torch.manual_seed(10)
            h = Hidden(HiDDenConfiguration(H=64, W=64), torch.device("cuda"), NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
            images = detgen.uniform((4, 3, 64, 64), 151).cuda(); messages = (detgen.uniform((4, 30), 152) > 0.5).float().cuda()
            losses, _ = h.train_on_batch([images, messages])
            outs.append((losses, [p.grad.detach().clone() for p in list(h.encoder_decoder.parameters()) + list(h.discriminator.parameters())]))
    finally:
        L.wm_debug_bwdst(ctypes.c_int(1))
    for k in outs[0][0]:
        assert abs(outs[0][0][k] - outs[1][0][k]) <= 1e-6 * max(1.0, abs(outs[1][0][k])), k
    worst = 0.0
    for a, b in zip(outs[0][1], outs[1][1]):
        den = b.abs().max().item()
        if den > 0:
            worst = max(worst, (a - b).abs().max().item() / den)
    # bf16 dy downstream of a coefficient that moved by 1e-7 can flip a rounding here and there
    assert worst < 2e-2, worst


@pytest.mark.parametrize("case", [(2, 40, 36, True), (1, 17, 50, False), (3, 64, 64, True), (17, 16, 16, True), (1, 33, 31, False)])
def test_dgrad_with_fused_bn_backward_apply_tensor_gradient(case):
    """wm_conv3x3_dgrad_applyfused (64 -> 64 layer, g a tensor): dy and dx bit-identical to wm_bn_bwd_apply followed by the plain
    dgrad; with the feeding layer's raw output also the same partial sums as wm_conv3x3_dgrad_bwdstats."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, feed = case
    C = 64
    g = nhwc(detgen.normal((B, C, H, W), 161), torch.bfloat16)
    y = nhwc(detgen.normal((B, C, H, W), 162, mean=0.3), torch.bfloat16)
    w = detgen.normal((C, C, 3, 3), 163, std=0.05).cuda()
    wpt = ops.pack_w3x3(w, C, C, torch.bfloat16, transpose=True)
    gamma = detgen.normal((C,), 164, mean=1.0, std=0.3).cuda(); beta = detgen.normal((C,), 165, std=0.3).cuda()
    yf = y.float()
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    scale = gamma * invstd
    stats = torch.stack([scale, beta - mean * scale, mean, invstd]).contiguous()
    ry = nhwc(detgen.normal((B, C, H, W), 166, mean=-0.2), torch.bfloat16)
    rsc = detgen.normal((C,), 167, mean=1.0, std=0.3).cuda(); rsh = detgen.normal((C,), 168, std=0.3).cuda()
    assert ops.conv3x3_dgrad_applyfused_supported(64, 64, torch.bfloat16)
    z = torch.zeros(C, device="cuda")
    coef = ops.bn_bwd_coef(g, None, y, stats, C, gamma, z.clone(), z.clone(), False)
    dy0 = ops.bn_bwd(g, None, y, stats, C, gamma, z.clone(), z.clone(), False, None, coef=coef)
    if feed:
        dx0, part0 = ops.conv3x3_dgrad_bwdstats(dy0, wpt, ry, rsc, rsh)
        dy1, dx1, part1 = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, ry, rsc, rsh)
        assert torch.equal(part0, part1)
    else:
        dx0, _ = ops.conv3x3_fwd(dy0, wpt, None, None, None, want_stats=False)
        dy1, dx1, part1 = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt)
        assert part1 is None
    assert dy0.float().abs().max().item() > 0 and dx0.float().abs().max().item() > 0
    assert torch.equal(dy0, dy1)
    assert torch.equal(dx0, dx1)
    # image-fed layer: the input gradient has 3 (-> 32) channels
    w3 = detgen.normal((C, 3, 3, 3), 169, std=0.05).cuda()
    wpt3 = ops.pack_w3x3(w3, C, 32, torch.bfloat16, transpose=True)
    dx2, _ = ops.conv3x3_fwd(dy0, wpt3, None, None, None, want_stats=False)
    dy3, dx3, _ = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt3)
    assert dx2.float().abs().max().item() > 0 and torch.equal(dy0, dy3) and torch.equal(dx2, dx3)


def test_fused_apply_keeps_the_training_step(debug_lib):
    """the HiDDeN step with the apply pass inside the dgrad kernels is bit-identical to the step with the stand-alone pass"""
    import ctypes
    from video_watermarking_forgery_detection_amd import _lib
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    L = debug_lib   # the -DWM_DEBUG build: the release library has no A/B switches
    outs = []
    try:
        L.wm_debug_bwd_fuse(ctypes.c_int(0))   # (the one-kernel backward sums over 8x16 tiles: compared at a tolerance in the test below)
        for on in (1, 0):
            L.wm_debug_apply_fuse(ctypes.c_int(on))
            torch.manual_seed(10)
            h = Hidden(HiDDenConfiguration(H=64, W=64), torch.device("cuda"), NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
            images = detgen.uniform((4, 3, 64, 64), 171).cuda(); messages = (detgen.uniform((4, 30), 172) > 0.5).float().cuda()
            for _ in range(2):
                losses, _ = h.train_on_batch([images, messages])
            outs.append((dict(losses), [p.detach().clone() for p in list(h.encoder_decoder.parameters()) + list(h.discriminator.parameters())]))
    finally:
        L.wm_debug_apply_fuse(ctypes.c_int(1))
        L.wm_debug_bwd_fuse(ctypes.c_int(1))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_one_kernel_backward_keeps_the_training_step(debug_lib):
    """the HiDDeN step with the body layers' backward in one kernel (csrc/bwd_ws.hip) against the two-kernel form: the same gradients up
    to the summation order of the weight gradient and the BatchNorm sums (f32), so the first step's parameters agree to f32 round-off
    through Adam"""
    import ctypes
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    L = debug_lib
    outs = []
    try:
        for on in (1, 0):
            L.wm_debug_bwd_fuse(ctypes.c_int(on))
            torch.manual_seed(10)
            h = Hidden(HiDDenConfiguration(H=64, W=64), torch.device("cuda"), NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
            images = detgen.uniform((4, 3, 64, 64), 171).cuda(); messages = (detgen.uniform((4, 30), 172) > 0.5).float().cuda()
            g = [p.detach().clone() for p in h.encoder_decoder.parameters()]
            losses, _ = h.train_on_batch([images, messages])
            outs.append((dict(losses), [p.detach().clone() for p in list(h.encoder_decoder.parameters()) + list(h.discriminator.parameters())],
                         h.enc_dec_flat_grad().clone() if hasattr(h, "enc_dec_flat_grad") else None))
    finally:
        L.wm_debug_bwd_fuse(ctypes.c_int(1))
    for k, v in outs[0][0].items():
        assert abs(v - outs[1][0][k]) <= 1e-5 * max(1.0, abs(v)), k       # the losses are computed before any backward
    worst = 0.0
    for a, b in zip(outs[0][1], outs[1][1]):
        worst = max(worst, (a - b).abs().max().item())
    assert worst <= 2.5e-3, worst      # one Adam step of lr 1e-3: a sign-like update flips where a gradient is at round-off level, nothing larger


@pytest.mark.parametrize("case", [(3, 40, 36, 64, torch.bfloat16), (2, 17, 50, 30, torch.bfloat16), (2, 32, 32, 64, torch.float32)])
def test_pooled_layer_bn_backward_sums_from_forward_pool_statistics(case):
    """wm_bnrelu_avgpool_stats (pooled mean + active-pixel count N+ and sum S+ per sample and channel) + wm_pooled_bn_bwd_rows +
    wm_bn_bwd_finalize_raw give the same pooled output, dgamma / dbeta / coef as wm_bnrelu_avgpool + wm_bn_bwd_reduce(gvec)."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, C, dt = case
    CP = 32 * ((C + 31) // 32)
    y = nhwc(detgen.normal((B, C, H, W), 181, mean=0.25), dt, CP)
    gamma = detgen.normal((C,), 182, mean=1.0, std=0.3).cuda(); beta = detgen.normal((C,), 183, std=0.3).cuda()
    yf = y.float()[..., :C]
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    stats = torch.zeros(4, CP, device="cuda")
    stats[0, :C] = gamma * invstd; stats[1, :C] = beta - mean * gamma * invstd; stats[2, :C] = mean; stats[3, :C] = invstd
    gvec = torch.zeros(B, CP, device="cuda"); gvec[:, :C] = detgen.normal((B, C), 184).cuda() / (H * W)
    p0 = ops.bnrelu_avgpool(y, stats[0], stats[1])
    p1, ps = ops.bnrelu_avgpool_stats(y, stats[0], stats[1])
    torch.testing.assert_close(p1, p0, rtol=1e-6, atol=1e-7)
    z = stats[0] * y.float() + stats[1]
    assert torch.equal(ps[0], (z > 0).float().sum((1, 2)))
    dg0 = torch.zeros(C, device="cuda"); db0 = torch.zeros(C, device="cuda")
    c0 = ops.bn_bwd_coef(None, gvec, y, stats, C, gamma, dg0, db0, False)
    dg1 = torch.full((C,), 0.5, device="cuda"); db1 = torch.full((C,), -0.5, device="cuda")
    c1 = ops.bn_bwd_coef_raw(ops.pooled_bwd_rows(gvec, ps), y, stats, C, gamma, dg1, db1, False)
    sc = dg0.abs().max().item() + db0.abs().max().item()
    torch.testing.assert_close(db1, db0, rtol=1e-4, atol=1e-5 * sc)
    torch.testing.assert_close(dg1, dg0, rtol=1e-4, atol=1e-5 * sc)
    torch.testing.assert_close(c1, c0, rtol=1e-4, atol=1e-5 * c0.abs().max().item())
    dg2 = torch.full((C,), 0.5, device="cuda"); db2 = torch.full((C,), -0.5, device="cuda")
    c2 = ops.bn_bwd_coef_pooled(gvec, ps, y, stats, C, gamma, dg2, db2, False)     # the two steps in one launch
    torch.testing.assert_close(c2, c1, rtol=1e-6, atol=1e-7 * c0.abs().max().item())
    torch.testing.assert_close(dg2, dg1, rtol=1e-6, atol=1e-7 * sc); torch.testing.assert_close(db2, db1, rtol=1e-6, atol=1e-7 * sc)


@pytest.mark.parametrize("case", [(2, 40, 36), (1, 64, 64), (3, 17, 50)])
def test_backward_sweep_hint_changes_only_the_summation_order(case):
    """sweep_reverse: a persistent conv / wgrad launch walks its tiles backwards (Infinity-Cache reuse along a
    chain of layers).  Outputs are bit-identical (the halo-edge reuse mirrors: right columns from the tile before);
    per-workgroup statistics rows and weight-gradient slabs only change their summation order."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W = case
    C = 64
    x = nhwc(detgen.normal((B, C, H, W), 191), torch.bfloat16)
    xs = detgen.normal((C,), 192, mean=1.0, std=0.3).cuda(); xt = detgen.normal((C,), 193, std=0.3).cuda()
    w = detgen.normal((C, C, 3, 3), 194, std=0.05).cuda()
    wp = ops.pack_w3x3(w, C, C, torch.bfloat16)
    y0, st0 = ops.conv3x3_fwd(x, wp, None, xs, xt, want_stats=True)
    y1, st1 = ops.conv3x3_fwd(x, wp, None, xs, xt, want_stats=True, reverse=True)
    y2, _ = ops.conv3x3_fwd(x, wp, None, xs, xt, want_stats=True)          # a per-call argument: nothing lingers
    assert y0.float().abs().max().item() > 0 and torch.equal(y0, y1) and torch.equal(y0, y2)
    torch.testing.assert_close(st1.sum(0), st0.sum(0), rtol=1e-5, atol=1e-3)
    dy = nhwc(detgen.normal((B, C, H, W), 195), torch.bfloat16)
    dw0 = torch.zeros(C, C, 3, 3, device="cuda"); dw1 = torch.zeros(C, C, 3, 3, device="cuda")
    ops.conv3x3_wgrad(x, C, xs, xt, dy, dw0, False)
    ops.conv3x3_wgrad(x, C, xs, xt, dy, dw1, False, reverse=True)
    torch.testing.assert_close(dw1, dw0, rtol=1e-5, atol=1e-5 * dw0.abs().max().item())
    g = nhwc(detgen.normal((B, C, H, W), 196), torch.bfloat16)
    gamma = detgen.normal((C,), 197, mean=1.0, std=0.3).cuda()
    yf = y0.float()
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    stats = torch.stack([gamma * invstd, -mean * gamma * invstd, mean, invstd]).contiguous()
    z = torch.zeros(C, device="cuda")
    coef = ops.bn_bwd_coef(g, None, y0, stats, C, gamma, z.clone(), z.clone(), False)
    wpt = ops.pack_w3x3(w, C, C, torch.bfloat16, transpose=True)
    a0 = ops.conv3x3_dgrad_applyfused(g, y0, stats, coef, wpt, x, xs, xt)
    a1 = ops.conv3x3_dgrad_applyfused(g, y0, stats, coef, wpt, x, xs, xt, reverse=True)
    assert torch.equal(a0[0], a1[0]) and torch.equal(a0[1], a1[1])
    torch.testing.assert_close(a1[2].sum(0), a0[2].sum(0), rtol=1e-5, atol=1e-3)


def test_body_layer_backward_at_the_baseline_size():
    """BASELINE size (B=16, 256x256, 64 -> 64, bf16): the fused backward of a body layer (apply pass + feeding layer's sums inside the
    input-gradient kernel, both sweep directions) is bit-identical to the stand-alone sequence, and the forward / input-gradient
    kernels satisfy the adjoint identity <conv(a), g> == <a, conv^T(g)> (a size-independent property, f32 accumulation)."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, C = 16, 256, 256, 64
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    g = torch.randn(B, H, W, C, device="cuda", generator=gen).to(torch.bfloat16)
    y = (torch.randn(B, H, W, C, device="cuda", generator=gen) + 0.3).to(torch.bfloat16)
    ry = (torch.randn(B, H, W, C, device="cuda", generator=gen) - 0.2).to(torch.bfloat16)
    w = detgen.normal((C, C, 3, 3), 221, std=0.05).cuda()
    wp = ops.pack_w3x3(w, C, C, torch.bfloat16)
    wpt = ops.pack_w3x3(w, C, C, torch.bfloat16, transpose=True)
    gamma = detgen.normal((C,), 222, mean=1.0, std=0.3).cuda(); beta = detgen.normal((C,), 223, std=0.3).cuda()
    yf = y.float()
    mean = yf.mean((0, 1, 2)); invstd = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
    stats = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()
    rsc = detgen.normal((C,), 224, mean=1.0, std=0.3).cuda(); rsh = detgen.normal((C,), 225, std=0.3).cuda()
    z = torch.zeros(C, device="cuda")
    coef = ops.bn_bwd_coef(g, None, y, stats, C, gamma, z.clone(), z.clone(), False)
    dy0 = ops.bn_bwd(g, None, y, stats, C, gamma, z.clone(), z.clone(), False, None, coef=coef)
    dx0, part0 = ops.conv3x3_dgrad_bwdstats(dy0, wpt, ry, rsc, rsh)
    for rev in (False, True):
        dy1, dx1, part1 = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, ry, rsc, rsh, reverse=rev)
        assert torch.equal(dy0, dy1) and torch.equal(dx0, dx1)
        torch.testing.assert_close(part1.sum(0), part0.sum(0), rtol=1e-5, atol=1e-2)
    # adjoint identity on the plain kernels: conv(a) with w, conv^T(gg) with the flipped / transposed w
    a = torch.randn(B, H, W, C, device="cuda", generator=gen).to(torch.bfloat16)
    gg = torch.randn(B, H, W, C, device="cuda", generator=gen).to(torch.bfloat16)
    ca, _ = ops.conv3x3_fwd(a, wp, None, None, None, want_stats=False)
    ctg, _ = ops.conv3x3_fwd(gg, wpt, None, None, None, want_stats=False, reverse=True)
    pl = ca.double() * gg.double(); pr = a.double() * ctg.double()
    lhs, rhs = pl.sum().item(), pr.sum().item()
    # each side rounds its 67M outputs to bf16 (relative error uniform in +-2^-9): the sums differ by a random walk of that size
    sigma = (2.0 ** -9 / 3 ** 0.5) * ((pl * pl).sum().item() ** 0.5 + (pr * pr).sum().item() ** 0.5)
    assert abs(lhs - rhs) <= 6 * sigma, (lhs, rhs, sigma)


@pytest.mark.parametrize("case", [(2, 40, 36, False), (16, 64, 64, True), (3, 33, 47, False)])
def test_bn_backward_finalisation_riding_on_the_weight_gradient_reduction(case):
    """wm_conv3x3_wgrad_fin / wm_conv3x3_wgrad_gvfused_fin: a few extra workgroups of the slab-reduction launch finish ANOTHER layer's
    BatchNorm-backward sums (dgamma, dbeta, coef) -- bit-identical to wm_bn_bwd_finalize_raw, and dw unchanged."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, pooled = case
    C = 64
    x = nhwc(detgen.normal((B, C, H, W), 231), torch.bfloat16)
    xs = detgen.normal((C,), 232, mean=1.0, std=0.3).cuda(); xt = detgen.normal((C,), 233, std=0.3).cuda()
    dy = nhwc(detgen.normal((B, C, H, W), 234), torch.bfloat16)
    # the other layer: partial rows as a dgrad epilogue would leave them
    n = ops.conv3x3_nparts(B, H, W, C, C, torch.bfloat16)
    part = detgen.normal((n, 2, C), 235).cuda().contiguous()
    yprev = nhwc(detgen.normal((B, C, H, W), 236, mean=0.3), torch.bfloat16)
    gamma = detgen.normal((C,), 237, mean=1.0, std=0.3).cuda()
    mean = detgen.normal((C,), 238, std=0.3).cuda(); invstd = detgen.uniform((C,), 239).cuda() + 0.5
    pstats = torch.stack([gamma * invstd, -mean * gamma * invstd, mean, invstd]).contiguous()
    for accumulate in (False, True):
        dg0 = torch.full((C,), 0.25, device="cuda"); db0 = torch.full((C,), -0.5, device="cuda")
        c0 = ops.bn_bwd_coef_raw(part.clone(), yprev, pstats, C, gamma, dg0, db0, accumulate)
        dg1 = torch.full((C,), 0.25, device="cuda"); db1 = torch.full((C,), -0.5, device="cuda")
        fin = dict(partials=part, y_shape=tuple(yprev.shape), stats=pstats, C=C, gamma=gamma, dgamma=dg1, dbeta=db1, accumulate=accumulate)
        dw0 = torch.zeros(C, C, 3, 3, device="cuda"); dw1 = torch.zeros(C, C, 3, 3, device="cuda")
        if pooled:
            gvec = detgen.normal((B, C), 240).cuda() / (H * W)
            yf = dy.float()
            m2 = yf.mean((0, 1, 2)); i2 = torch.rsqrt(yf.var((0, 1, 2), unbiased=False) + 1e-5)
            st = torch.stack([i2, -m2 * i2, m2, i2]).contiguous()
            z = torch.zeros(C, device="cuda")
            coef = ops.bn_bwd_coef(None, gvec, dy, st, C, torch.ones(C, device="cuda"), z.clone(), z.clone(), False)
            ops.conv3x3_wgrad_gvfused(x, xs, xt, gvec, dy, st, coef, dw0, False)
            c1 = ops.conv3x3_wgrad_gvfused(x, xs, xt, gvec, dy, st, coef, dw1, False, fin=fin)
        else:
            assert ops.conv3x3_wgrad(x, C, xs, xt, dy, dw0, False) is None
            c1 = ops.conv3x3_wgrad(x, C, xs, xt, dy, dw1, False, fin=fin)
        assert dw0.abs().max().item() > 0 and torch.equal(dw0, dw1)
        assert torch.equal(c0, c1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)


def test_finalisation_riders_keep_the_training_step(debug_lib):
    """the HiDDeN step with the feeding layers' BatchNorm-backward finalisations riding on the weight-gradient reductions is
    bit-identical to the step with stand-alone finalisation launches"""
    import ctypes
    from video_watermarking_forgery_detection_amd import _lib
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    L = debug_lib   # the -DWM_DEBUG build: the release library has no A/B switches
    outs = []
    try:
        for on in (1, 0):
            L.wm_debug_fin_rider(ctypes.c_int(on))
            torch.manual_seed(10)
            h = Hidden(HiDDenConfiguration(H=64, W=64), torch.device("cuda"), NL.Jpeg(50), None, compute_dtype=torch.bfloat16)
            images = detgen.uniform((4, 3, 64, 64), 241).cuda(); messages = (detgen.uniform((4, 30), 242) > 0.5).float().cuda()
            for _ in range(3):
                losses, _ = h.train_on_batch([images, messages])
            outs.append((dict(losses), [p.detach().clone() for p in list(h.encoder_decoder.parameters()) + list(h.discriminator.parameters())]))
    finally:
        L.wm_debug_fin_rider(ctypes.c_int(1))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_fused_backward_entry_points_fail_loudly():
    """the fused backward entry points reject what they do not support with a negative code and a message (no silent fallback):
    wrong dtype, wrong channel counts, missing partner arguments, an oversized rider"""
    import ctypes
    from video_watermarking_forgery_detection_amd import _lib, ops
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    B, H, W, C = 1, 16, 16, 64
    g = torch.zeros(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    f = torch.zeros(4, C, device="cuda"); coef = torch.zeros(3, C, device="cuda")
    wpt = torch.zeros(9, C, C, device="cuda", dtype=torch.bfloat16)
    part = torch.zeros(1, 2, C, device="cuda")
    def err():
        return L.wm_last_error_string().decode()
    # f32 has no fused kernel
    assert not ops.conv3x3_dgrad_applyfused_supported(64, 64, torch.float32) and not ops.conv3x3_gvfused_supported(64, 64, torch.float32)
    rc = L.wm_conv3x3_dgrad_applyfused(P(g), P(g), P(f), P(coef), P(wpt), P(g), P(g), P(None), P(None), P(None), P(None), 1, 16, 16, 64, 0, None)
    assert rc < 0 and "unsupported" in err()
    # the feeding layer's sums need all four of ry, r_scale, r_shift, partials
    rc = L.wm_conv3x3_dgrad_applyfused(P(g), P(g), P(f), P(coef), P(wpt), P(g), P(g), P(g), P(None), P(None), P(part), 1, 16, 16, 64, 1, None)
    assert rc < 0 and "come together" in err()
    # ... and a 64-channel input gradient
    rc = L.wm_conv3x3_dgrad_applyfused(P(g), P(g), P(f), P(coef), P(wpt), P(g), P(g), P(g), P(f), P(f), P(part), 1, 16, 16, 32, 1, None)
    assert rc < 0 and "CinP" in err()
    rc = L.wm_conv3x3_dgrad_bwdstats(P(g), 64, 48, P(wpt), P(None), P(None), P(None), P(g), P(f), P(f), P(g), P(part), 1, 16, 16, 64, 1, None)
    assert rc < 0 and "unsupported" in err()
    # a rider with more than 256 partial rows is refused by the weight-gradient entry (python wrapper: assertion; C ABI: code)
    big = torch.zeros(300, 2, C, device="cuda")
    fin = dict(partials=big, y_shape=(B, H, W, C), stats=f, C=C, gamma=f[0], dgamma=f[1].clone(), dbeta=f[2].clone(), accumulate=False)
    with pytest.raises(AssertionError):
        ops.conv3x3_wgrad(g, C, f[0], f[1], g, torch.zeros(C, C, 3, 3, device="cuda"), False, fin=fin)
    st = ops._WmBnBwdFin(partials=big.data_ptr(), nparts=300, C=C, CP=C, count=256.0, gamma=f[0].data_ptr(), mean=f[2].data_ptr(),
                         invstd=f[3].data_ptr(), dgamma=0, dbeta=0, accumulate=0, coef=coef.data_ptr())
    ws = torch.zeros(ops._lib.lib().wm_conv3x3_wgrad_nslabs(1, 16, 16) * 9 * 64 * 64, device="cuda")
    dw = torch.zeros(C, C, 3, 3, device="cuda")
    rc = L.wm_conv3x3_wgrad_fin(P(g), 64, 64, P(f[0]), P(f[1]), P(g), 64, 64, P(ws), P(dw), 0, 1, 16, 16, 64, 64, P(None), 1, ctypes.byref(st), None)
    assert rc < 0 and "rider" in err()
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 40, 36), (1, 16, 16), (3, 33, 18)])
def test_after_concat_layer_without_the_concat(case, dtype):
    """hidden_models/encoder.py:25,34-41 in its split form -- conv64(features) + [conv3(image) + bias + message term] -- against
    torch's CPU conv over the materialised 97-channel concat (operands rounded to the 16-bit dtype the kernels store): forward
    (side tensor P, the summed output, its statistics) and the three slices of the weight gradient (message / feature / image channels)"""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W = case
    L, C = 30, 64
    img = detgen.uniform((B, 3, H, W), 301)
    msg = detgen.bits((B, L), 302)
    feat = detgen.normal((B, C, H, W), 303).to(dtype).float()
    sc = detgen.normal((C,), 304, mean=1.0, std=0.2); sh = detgen.normal((C,), 305, std=0.3)
    w = detgen.normal((C, L + C + 3, 3, 3), 306, std=(2.0 / (9 * 97)) ** 0.5)
    bias = detgen.normal((C,), 307, std=0.1)
    wq = w.clone()
    wq[:, L:] = w[:, L:].to(dtype).float()         # feature and image filters are stored in the 16-bit dtype; the message term stays f32
    a = torch.relu(feat * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dtype).float()
    imq = img.to(dtype).float()
    cat = torch.cat([msg.view(B, L, 1, 1).expand(-1, -1, H, W), a, imq], dim=1)
    wr = wq.clone().requires_grad_(True)
    ref = F.conv2d(cat, wr, bias, padding=1)
    dy = detgen.normal((B, C, H, W), 308).to(dtype).float()
    ref.backward(dy)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    # ---- forward
    P = ops.concat_side_fwd(img.cuda(), w.cuda(), bias.cuda(), msg.cuda(), dtype, 0, L, L + C)
    ref_side = F.conv2d(cat[:, :L], wq[:, :L], bias, padding=1) + F.conv2d(imq, wq[:, L + C:], None, padding=1)
    torch.testing.assert_close(nchw(P, C), ref_side, rtol=tol, atol=tol)
    DROP = 1 << 20
    fperm = [DROP] * L + list(range(C)) + [DROP] * 3
    wp = ops.pack_w3x3(w.cuda(), C, C, dtype, perm=fperm)
    y, st = ops.conv3x3_fwd_addin(nhwc(feat, dtype), wp, sc.cuda(), sh.cuda(), P)
    torch.testing.assert_close(nchw(y, C), ref.detach(), rtol=tol, atol=2 * tol)
    s = st.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.detach().sum((0, 2, 3)), rtol=2e-2, atol=0.5)
    # ---- weight gradient: three writers, three disjoint channel ranges of dW
    dw = torch.full((C, L + C + 3, 3, 3), 7.0, device="cuda")
    dyh = nhwc(dy, dtype)
    ops.conv3x3_wgrad(nhwc(feat, dtype), C, sc.cuda(), sh.cuda(), dyh, dw, False, perm_dev=torch.tensor(fperm, dtype=torch.int32, device="cuda"))
    img16 = torch.zeros(B, H, W, 16, device="cuda", dtype=dtype)
    ops.nchw_to_nhwc(img.cuda(), img16, 0, 13)
    iperm = [DROP] * (L + C) + [0, 1, 2]
    ops.conv3x3_wgrad(img16, 16, None, None, dyh, dw, False, perm_dev=torch.tensor(iperm, dtype=torch.int32, device="cuda"))
    ops.concat_side_msg_wgrad(dyh, msg.cuda(), dw, False, 0, L)
    g = wr.grad
    torch.testing.assert_close(dw.cpu(), g, rtol=tol, atol=tol * g.abs().max().item())
    # accumulate adds onto what is there
    dw2 = dw.clone()
    ops.concat_side_msg_wgrad(dyh, msg.cuda(), dw2, True, 0, L)
    torch.testing.assert_close(dw2[:, :L].cpu(), 2 * dw[:, :L].cpu(), rtol=1e-5, atol=1e-5 * g.abs().max().item())
    assert torch.equal(dw2[:, L:], dw[:, L:])


@pytest.mark.parametrize("case", [(2, 64, 64, torch.bfloat16, False), (3, 40, 56, torch.bfloat16, True), (1, 21, 37, torch.float16, False),
                                  (16, 128, 128, torch.bfloat16, True),
                                  (2, 16, 16, torch.bfloat16, False), (5, 8, 200, torch.bfloat16, True)])   # one tile column / one tile row (the mulhi geometry's d = 1)
def test_fused_backward_kernel_against_the_two_kernel_form(case):
    """csrc/bwd_ws.hip (input gradient + the feeding layer's BatchNorm sums + weight gradient from one staged dy / a tile) against
    wm_conv3x3_dgrad_applyfused + wm_conv3x3_wgrad on the same operands: dx bit-identical (same dy, same MFMA order over K), the
    sums and the weight gradient equal up to the summation order over tiles."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, dt, rev = case
    C = 64
    g = nhwc(detgen.normal((B, C, H, W), 901), dt, C)
    y = nhwc(detgen.normal((B, C, H, W), 902, mean=0.2), dt, C)
    xr = nhwc(detgen.normal((B, C, H, W), 903, mean=0.1), dt, C)
    stats = torch.empty(4, C, device="cuda")
    stats[0] = detgen.normal((C,), 904, mean=1.0, std=0.3).cuda(); stats[1] = detgen.normal((C,), 905, std=0.3).cuda()
    stats[2] = detgen.normal((C,), 906, std=0.2).cuda(); stats[3] = detgen.uniform((C,), 907).cuda() + 0.5
    coef = torch.empty(3, C, device="cuda")
    coef[0] = detgen.normal((C,), 908, mean=1.0, std=0.2).cuda(); coef[1] = detgen.normal((C,), 909, std=0.01).cuda(); coef[2] = detgen.normal((C,), 910, std=0.01).cuda()
    in_scale = detgen.normal((C,), 911, mean=1.0, std=0.3).cuda(); in_shift = detgen.normal((C,), 912, std=0.3).cuda()
    w = detgen.normal((C, C, 3, 3), 913, std=0.05).cuda()
    wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
    dw0 = torch.zeros(C, C, 3, 3, device="cuda"); dw1 = torch.full((C, C, 3, 3), 0.25, device="cuda")
    dy, dx0, part0 = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, xr, in_scale, in_shift, reverse=rev)
    ops.conv3x3_wgrad(xr, C, in_scale, in_shift, dy, dw0, False, reverse=not rev)
    dx1, part1, _ = ops.conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, in_scale, in_shift, dw1, True, reverse=rev)
    torch.cuda.synchronize()
    assert dx0.float().abs().max().item() > 0
    # the one-kernel form writes dx already multiplied by the feeding layer's ReLU mask (its consumers apply the mask anyway)
    z = in_scale * xr.float() + in_shift
    sure = z.abs() > 1e-4                       # (the kernel evaluates z with one fma: the sign may differ within round-off of zero)
    dx0m = torch.where(z > 0, dx0, torch.zeros_like(dx0))
    if dt == torch.bfloat16:
        assert torch.equal(dx0m[sure], dx1[sure])
    else:   # f16: one dy element in ~50,000 lands on the other side of a rounding tie (an f32 last-bit difference in the folded constants'
        # evaluation order); its 3x3 neighbourhood of dx then differs by one f16 ulp here and there
        diff = (dx0m.float() - dx1.float()).abs()[sure]
        assert (diff > 0).float().mean().item() < 2e-3 and diff.max().item() <= 2.0 ** -10 * max(1.0, dx0.float().abs().max().item())
    # a gradient that arrives already masked by this layer's ReLU (what this kernel itself emits) gives the same results with the flag
    zl = stats[0] * y.float() + stats[1]
    gm = torch.where(zl > 0, g, torch.zeros_like(g))
    dw2 = torch.full((C, C, 3, 3), 0.25, device="cuda")
    dx2, part2, _ = ops.conv3x3_bwd_fused(gm, y, stats, coef, wpt, xr, in_scale, in_shift, dw2, True, reverse=rev, premasked=True)
    dw3 = torch.full((C, C, 3, 3), 0.25, device="cuda")
    dx3, part3, _ = ops.conv3x3_bwd_fused(gm, y, stats, coef, wpt, xr, in_scale, in_shift, dw3, True, reverse=rev, premasked=False)
    assert torch.equal(dx2, dx3) and torch.equal(part2, part3) and torch.equal(dw2, dw3)
    near0 = (zl.abs() <= 1e-4).float().mean().item()
    assert near0 < 1e-3 and (dx2.float() - dx1.float()).abs().max().item() <= (1e-6 if near0 == 0 else 1.0)   # = the unmasked g's result unless a z sits at zero
    s0, s1 = part0.double().sum(0), part1.double().sum(0)
    assert (s0 - s1).abs().max().item() <= (1e-5 if dt == torch.bfloat16 else 1e-3) * s0.abs().max().item()
    ref = dw0 + 0.25
    assert (dw1 - ref).abs().max().item() <= 2e-4 * dw0.abs().max().item()


@pytest.mark.parametrize("case", [(4, 64, 64, torch.bfloat16), (4, 64, 64, torch.float16), (3, 40, 56, torch.float16), (16, 128, 128, torch.bfloat16)])
def test_fused_backward_kernel_gvec_form(case):
    """the one-kernel backward of a globally pooled layer (one gradient row per sample) against wm_conv3x3_dgrad_bwdstats(gvec) +
    wm_conv3x3_wgrad_gvfused"""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, dt = case
    C = 64
    y = nhwc(detgen.normal((B, C, H, W), 922, mean=0.2), dt, C)
    xr = nhwc(detgen.normal((B, C, H, W), 923, mean=0.1), dt, C)
    gvec = (detgen.normal((B, C), 921) / (H * W)).cuda().contiguous()
    stats = torch.empty(4, C, device="cuda")
    stats[0] = detgen.normal((C,), 904, mean=1.0, std=0.3).cuda(); stats[1] = detgen.normal((C,), 905, std=0.3).cuda()
    stats[2] = detgen.normal((C,), 906, std=0.2).cuda(); stats[3] = detgen.uniform((C,), 907).cuda() + 0.5
    coef = torch.empty(3, C, device="cuda")
    coef[0] = detgen.normal((C,), 908, mean=1.0, std=0.2).cuda(); coef[1] = detgen.normal((C,), 909, std=1e-5).cuda(); coef[2] = detgen.normal((C,), 910, std=1e-5).cuda()
    in_scale = detgen.normal((C,), 911, mean=1.0, std=0.3).cuda(); in_shift = detgen.normal((C,), 912, std=0.3).cuda()
    w = detgen.normal((C, C, 3, 3), 913, std=0.05).cuda()
    wpt = ops.pack_w3x3(w, C, C, dt, transpose=True)
    dw0 = torch.zeros(C, C, 3, 3, device="cuda"); dw1 = torch.zeros(C, C, 3, 3, device="cuda")
    dx0, part0 = ops.conv3x3_dgrad_bwdstats(y, wpt, xr, in_scale, in_shift, gvec, stats, coef)
    ops.conv3x3_wgrad_gvfused(xr, in_scale, in_shift, gvec, y, stats, coef, dw0, False)
    dx1, part1, _ = ops.conv3x3_bwd_fused(None, y, stats, coef, wpt, xr, in_scale, in_shift, dw1, False, gvec=gvec)
    torch.cuda.synchronize()
    assert dx0.float().abs().max().item() > 0
    z = in_scale * xr.float() + in_shift
    sure = z.abs() > 1e-4
    dx0m = torch.where(z > 0, dx0, torch.zeros_like(dx0))
    diff = (dx0m.float() - dx1.float()).abs()[sure]
    if dt == torch.bfloat16:
        assert diff.max().item() == 0.0
    else:
        assert (diff > 0).float().mean().item() < 2e-3
    s0, s1 = part0.double().sum(0), part1.double().sum(0)
    assert (s0 - s1).abs().max().item() <= (1e-5 if dt == torch.bfloat16 else 1e-3) * s0.abs().max().item()
    assert (dw1 - dw0).abs().max().item() <= 2e-4 * dw0.abs().max().item()
    with pytest.raises(RuntimeError):
        ops.conv3x3_bwd_fused(None, y, stats, coef, wpt, xr, in_scale, in_shift, dw1, False)      # neither g nor gvec


@pytest.mark.parametrize("case", [(2, 64, 64, torch.bfloat16, False), (3, 24, 48, torch.bfloat16, True), (1, 8, 16, torch.bfloat16, False),
                                  (16, 128, 128, torch.bfloat16, True), (2, 40, 32, torch.float16, False), (5, 8, 208, torch.bfloat16, True)])
def test_fused_backward_of_an_image_fed_first_layer(case):
    """csrc/bwd_ws16.hip (3 -> 64 first layer whose image input needs a gradient: dx and dW from one staged dy tile) against the two-kernel
    form on the same operands: wm_conv3x3_dgrad_applyfused (writes dy, dx with 32 channels) + wm_conv3x3_wgrad on the 16-channel image."""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, dt, rev = case
    C = 64
    g = nhwc(detgen.normal((B, C, H, W), 951, std=0.1), dt, C)
    y = nhwc(detgen.normal((B, C, H, W), 952, mean=0.2), dt, C)
    x16 = torch.zeros(B, H, W, 16, device="cuda", dtype=dt)
    ops.nchw_to_nhwc(detgen.uniform((B, 3, H, W), 953).cuda(), x16, 0, 13)
    stats = torch.empty(4, C, device="cuda")
    stats[0] = detgen.normal((C,), 954, mean=1.0, std=0.3).cuda(); stats[1] = detgen.normal((C,), 955, std=0.3).cuda()
    stats[2] = detgen.normal((C,), 956, std=0.2).cuda(); stats[3] = detgen.uniform((C,), 957).cuda() + 0.5
    coef = torch.empty(3, C, device="cuda")
    coef[0] = detgen.normal((C,), 958, mean=1.0, std=0.2).cuda(); coef[1] = detgen.normal((C,), 959, std=0.01).cuda(); coef[2] = detgen.normal((C,), 960, std=0.01).cuda()
    w = detgen.normal((C, 3, 3, 3), 961, std=0.2).cuda()
    assert ops.conv3x3_bwd_fused16_supported(y.shape, dt) and not ops.conv3x3_bwd_fused16_supported((B, H + 1, W, C), dt)
    # the two-kernel form
    wpt32 = ops.pack_w3x3(w, C, 32, dt, transpose=True)
    dy, dx0, _ = ops.conv3x3_dgrad_applyfused(g, y, stats, coef, wpt32, reverse=rev)
    dw0 = torch.zeros(C, 3, 3, 3, device="cuda")
    ops.conv3x3_wgrad(x16, 16, None, None, dy, dw0, False, reverse=not rev)
    # one pass
    wpt16 = ops.pack_w3x3(w, C, 16, dt, transpose=True)
    dw1 = torch.full((C, 3, 3, 3), 0.25, device="cuda")
    dx1 = ops.conv3x3_bwd_fused16(g, y, stats, coef, wpt16, x16, dw1, True, reverse=rev)
    torch.cuda.synchronize()
    assert tuple(dx1.shape) == (B, H, W, 16) and dx0.float().abs().max().item() > 0
    assert float(dx1[..., 3:].float().abs().max()) == 0.0                       # the 13 padding channels of the image tensor
    if dt == torch.bfloat16:
        assert torch.equal(dx1[..., :3], dx0[..., :3])                          # same dy, same MFMA order over K
    else:
        d = (dx1[..., :3].float() - dx0[..., :3].float()).abs()
        assert (d > 0).float().mean().item() < 2e-3 and d.max().item() <= 2.0 ** -10 * max(1.0, dx0.float().abs().max().item())
    assert (dw1 - (dw0 + 0.25)).abs().max().item() <= 2e-4 * dw0.abs().max().item()
    # a gradient already multiplied by the layer's ReLU mask gives the same results with the flag
    zl = stats[0] * y.float() + stats[1]
    gm = torch.where(zl > 0, g, torch.zeros_like(g))
    dw2 = torch.zeros(C, 3, 3, 3, device="cuda"); dw3 = torch.zeros(C, 3, 3, 3, device="cuda")
    dx2 = ops.conv3x3_bwd_fused16(gm, y, stats, coef, wpt16, x16, dw2, False, reverse=rev, premasked=True)
    dx3 = ops.conv3x3_bwd_fused16(gm, y, stats, coef, wpt16, x16, dw3, False, reverse=rev, premasked=False)
    assert torch.equal(dx2, dx3) and torch.equal(dw2, dw3)
    near0 = (zl.abs() <= 1e-4).float().mean().item()
    assert near0 < 1e-3 and (dx2.float() - dx1.float()).abs().max().item() <= (1e-6 if near0 == 0 else 1.0)
