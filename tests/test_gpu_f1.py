"""GPU: SURVEY 8f row 1 -- networks.Discriminator, FBCNN, QF_predictor on the general HIP layer family (csrc/gconv.hip, gelem.hip).

  * per kernel, through the C ABI: conv forward / input gradient / weight gradient / transposed conv for every kernel size, stride
    and padding the three networks use (and ragged channel counts), against torch's CPU conv; f32 on the exact-f32 MFMA at 1e-4,
    bf16 / f16 at their mantissa bounds; the activations, QF-attention combine, global pool, symmetric / replication padding,
    spectral norm (forward, power iteration, backward) and the Bayar constraint against the oracle's definitions;
  * per network, f32: outputs, input gradients, every parameter gradient and the power-iteration state against tests/golden/f1.npz,
    which the REFERENCE's classes generated (make_golden.py gen_f1);
  * per network, bf16 and f16: against the CPU oracle at other sizes, at the 16-bit bound;
  * the surface: state_dict round trip from a reference-keyed dictionary, CPU tensors and wrong shapes refused loudly.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detgen
from oracle import f1_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def rel2(a, b):
    """relative L2 error: the measure for gradients that went back through a deep 16-bit ReLU stack (single mask flips dominate the
    max-norm)"""
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-12))


TOL = {torch.float32: 1e-4, torch.bfloat16: 2.5e-2, torch.float16: 4e-3}


def nhwc(x, dtype):
    from video_watermarking_forgery_detection_amd import ops
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, ops.cpad(C), dtype=dtype, device=DEV)
    out[..., :C] = x.permute(0, 2, 3, 1).to(DEV).to(dtype)
    return out.contiguous()


def nchw(x, C):
    return x[..., :C].permute(0, 3, 1, 2).float().cpu()


GEOS = [  # (Cin, Cout, k, stride, pad, H, W)   -- every geometry of the three networks, plus ragged ones
    (3, 32, 4, 2, 1, 32, 32), (32, 32, 3, 1, 1, 16, 16), (64, 128, 4, 2, 1, 16, 16), (512, 1, 1, 1, 0, 2, 2),
    (3, 16, 3, 1, 1, 20, 24), (16, 32, 2, 2, 0, 20, 24), (48, 48, 3, 1, 1, 5, 6), (3, 3, 5, 1, 0, 36, 36),
    (48, 192, 2, 2, 0, 8, 8), (192, 192, 3, 1, 1, 4, 4), (5, 7, 5, 2, 2, 13, 11), (20, 40, 3, 2, 1, 9, 9), (1, 512, 1, 1, 0, 1, 1),
    (3, 16, 5, 1, 2, 6, 150), (3, 32, 4, 2, 1, 8, 260), (64, 32, 4, 2, 1, 6, 10),   # rows wider than a 64-pixel chunk (thin-layer weight gradient)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv_family_against_torch(dtype):
    from video_watermarking_forgery_detection_amd import ops
    for gi, (Cin, Cout, k, s, p, H, W) in enumerate(GEOS):
        B = 2
        x = detgen.normal((B, Cin, H, W), 100 + gi)
        w = detgen.normal((Cout, Cin, k, k), 200 + gi, std=(1.0 / (Cin * k * k)) ** 0.5)
        b = detgen.normal((Cout,), 300 + gi, std=0.1)
        xr, wr = x.to(dtype).float().clone(), w.to(dtype).float().clone()
        xr.requires_grad_(True); wr.requires_grad_(True); bb = b.clone().requires_grad_(True)
        y = F.conv2d(xr, wr, bb, stride=s, padding=p)
        g = detgen.normal(tuple(y.shape), 400 + gi)
        gr = g.to(dtype).float()
        (y * gr).sum().backward()
        OH, OW = y.shape[-2:]
        xh, gh = nhwc(x, dtype), nhwc(g, dtype)
        KC, NC = xh.shape[3], gh.shape[3]
        wp = ops.gconv_pack(w.to(DEV), NC, KC, False, dtype)
        bp = torch.zeros(NC, device=DEV); bp[:Cout] = b.to(DEV)
        yh = ops.gconv_fwd(xh, wp, bp, (OH, OW), k, k, s, p)
        tag = f"geo {gi} {dtype}"
        assert rel(nchw(yh, Cout), y) < TOL[dtype], tag
        assert float(yh[..., Cout:].abs().max() if NC > Cout else 0) == 0.0, tag           # padding channels stay zero
        wt = ops.gconv_pack(w.to(DEV), KC, NC, True, dtype)
        gxh = ops.gconv_fwd(gh, wt, None, (H, W), k, k, s, p, dgrad=True)
        assert rel(nchw(gxh, Cin), xr.grad) < TOL[dtype], tag
        dw, db = ops.gconv_wgrad(gh, xh, Cout, Cin, k, k, s, p)
        assert rel(dw, wr.grad) < TOL[dtype], tag
        assert rel(db, bb.grad) < TOL[dtype], tag


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transposed_conv_and_linear_layers(dtype):
    from video_watermarking_forgery_detection_amd import glayers as G
    for (Cin, Cout, H, W) in ((48, 48, 5, 6), (48, 32, 10, 12), (32, 16, 3, 3)):
        m = G.ConvTranspose2d(Cin, Cout, 2, 2, 0).to(DEV)
        x = detgen.normal((2, Cin, H, W), Cin + H)
        xr = x.to(dtype).float().clone().requires_grad_(True)
        wr = m.weight.detach().cpu().to(dtype).float().clone().requires_grad_(True)
        br = m.bias.detach().cpu().clone().requires_grad_(True)
        y = F.conv_transpose2d(xr, wr, br, stride=2)
        g = detgen.normal(tuple(y.shape), 77)
        (y * g.to(dtype).float()).sum().backward()
        xd = x.to(DEV).requires_grad_(True)
        yh = G.to_nchw(m(G.to_nhwc(xd, dtype)), Cout)
        (yh * g.to(dtype).float().to(DEV)).sum().backward()
        assert rel(yh, y) < TOL[dtype]
        assert rel(xd.grad, xr.grad) < TOL[dtype]
        assert rel(m.weight.grad, wr.grad) < TOL[dtype]
        assert rel(m.bias.grad, br.grad) < TOL[dtype]
    lin = G.Linear(37, 21).to(DEV)
    v = detgen.normal((5, 37), 5).to(DEV).requires_grad_(True)
    out = G.vector_out(lin(G.vector_in(v)), 21)
    ref = F.linear(v.detach().cpu(), lin.weight.detach().cpu(), lin.bias.detach().cpu())
    assert rel(out, ref) < 1e-5
    out.sum().backward()
    assert rel(v.grad, lin.weight.detach().sum(0).expand(5, -1)) < 1e-5
    assert rel(lin.bias.grad, torch.full((21,), 5.0)) < 1e-6


def test_activations_combines_pool():
    from video_watermarking_forgery_detection_amd import ops
    x = detgen.normal((2, 5, 6, 32), 1, std=2.0)
    g = detgen.normal((2, 5, 6, 32), 2)
    fns = {"relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2), "gelu": F.gelu, "elu": F.elu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}
    for kind, fn in fns.items():
        xr = x.clone().requires_grad_(True)
        y = fn(xr)
        (y * g).sum().backward()
        assert rel(ops.unary_fwd(x.to(DEV), kind), y) < 2e-6, kind
        assert rel(ops.unary_bwd(x.to(DEV), g.to(DEV), kind), xr.grad) < 5e-6, kind
        yb = ops.unary_fwd(x.to(DEV).bfloat16(), kind)
        assert rel(yb, fn(x.bfloat16().float())) < 1e-2, kind
    a, b = x.to(DEV), g.to(DEV)
    assert rel(ops.add_scaled(a, b, 0.5), x + 0.5 * g) < 1e-6
    gam, bet = detgen.uniform((2, 32), 3), detgen.normal((2, 32), 4)
    ref = x + gam[:, None, None, :] * g + bet[:, None, None, :]
    assert rel(ops.qfatt_fwd(a, b, gam.to(DEV), bet.to(DEV)), ref) < 1e-6
    gres, gg, gb = ops.qfatt_bwd(a, b, gam.to(DEV))            # upstream gradient a, res b
    assert rel(gres, gam[:, None, None, :] * x) < 1e-6
    assert rel(gg, (x * g).sum((1, 2))) < 1e-5
    assert rel(gb, x.sum((1, 2))) < 1e-5
    assert rel(ops.gpool_fwd(a), x.mean((1, 2))) < 1e-6
    assert rel(ops.gpool_bwd(gam.to(DEV), (2, 5, 6, 32), torch.float32), (gam / 30.0)[:, None, None, :].expand(2, 5, 6, 32)) < 1e-6


def test_activation_backward_with_the_bias_gradient():
    """ops.unary_bwd_colsum (the activation's backward and the split partials of the bias gradient in one pass) against unary_bwd + gcolsum, and a
    FusedSequential against the same children run one by one"""
    from video_watermarking_forgery_detection_amd import ops, glayers as G
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        for (B, H, W, C, creal, kind) in ((2, 9, 7, 16, 3, "elu"), (3, 33, 40, 64, 64, "relu"), (2, 64, 64, 192, 180, "gelu"), (8, 128, 128, 32, 32, "lrelu")):
            x = detgen.normal((B, H, W, C), 7 + C).to(DEV).to(dtype)
            g = detgen.normal((B, H, W, C), 8 + C).to(DEV).to(dtype)
            gx0 = ops.unary_bwd(x, g, kind)
            db0 = ops.gcolsum(gx0, creal)
            for _ in range(2):
                gx, db = ops.unary_bwd_colsum(x, g, kind, creal)
                assert rel(gx, gx0) <= {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype], (dtype, C, kind)   # (gelu's FMA contraction may differ by an ulp between the two kernels)
                assert rel(db, db0) < 1e-4, (dtype, C)      # (f32 sums of +- terms in two different orders)
    torch.manual_seed(0)
    seq = G.FusedSequential(G.Conv2d(5, 24, 3, 1, 1), G.Act("elu"), G.Conv2d(24, 24, 4, 2, 1), G.Act("relu"), G.Conv2d(24, 7, 1, 1, 0)).to(DEV)
    x = detgen.normal((2, 5, 12, 12), 3).to(DEV)
    for dtype in (torch.float32, torch.bfloat16):
        outs = []
        for fused in (True, False):
            seq.zero_grad()
            xd = x.clone().requires_grad_(True)
            h = G.to_nhwc(xd, dtype)
            if fused:
                h = seq(h)
            else:
                for m in seq:
                    h = m(h)
            y = G.to_nchw(h, 7)
            (y * y).sum().backward()
            outs.append([y.detach().clone(), xd.grad.clone()] + [p.grad.clone() for p in seq.parameters()])
        for a, b in zip(*outs):
            assert rel(a, b) < (1e-5 if dtype == torch.float32 else 1e-2)


def test_padding_layout_changes():
    from video_watermarking_forgery_detection_amd import glayers as G, ops
    from video_watermarking_forgery_detection_amd.models.conditional_jpeg_generator import symm_pad
    for (shape, pads) in (((1, 2, 5, 7), (2, 3, 4, 1)), ((2, 3, 32, 32), (2, 2, 2, 2)), ((1, 3, 3, 4), (5, 6, 7, 4))):
        l, r, t, b = pads
        im = detgen.uniform(shape, 9400)
        for mode, ref_fn in ((ops.PAD_SYMMETRIC, lambda z: f1_ref.symm_pad(z, pads)),
                             (ops.PAD_REPLICATE, lambda z: F.pad(z, (l, r, t, b), mode="replicate"))):
            zr = im.clone().requires_grad_(True)
            ref = ref_fn(zr)
            gy = detgen.normal(tuple(ref.shape), 11)
            (ref * gy).sum().backward()
            zd = im.to(DEV).requires_grad_(True)
            out = G.to_nchw(G.to_nhwc(zd, torch.float32, pads, mode), shape[1])
            assert torch.equal(out.cpu(), ref.detach()), (shape, pads, mode)
            (out * gy.to(DEV)).sum().backward()
            assert rel(zd.grad, zr.grad) < 1e-6, (shape, pads, mode)
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "f1.npz"))
    assert np.array_equal(symm_pad(detgen.uniform((1, 2, 5, 7), 9400).to(DEV), (2, 3, 4, 1)).cpu().numpy(), g["sympad/y"])
    # the window / crop on the way out, and its adjoint
    x = detgen.normal((2, 6, 9, 16), 5).to(DEV).requires_grad_(True)
    y = G.to_nchw(x, 3, 4, 7)
    assert torch.equal(y, x.detach()[:, :4, :7, :3].permute(0, 3, 1, 2))
    y.sum().backward()
    assert float(x.grad.sum()) == 2 * 4 * 7 * 3 and float(x.grad[:, 4:].abs().sum()) == 0


def test_spectral_norm_and_bayar():
    from video_watermarking_forgery_detection_amd import ops
    for (co, ci, k) in ((32, 3, 4), (64, 64, 3), (512, 256, 4)):
        w = detgen.normal((co, ci, k, k), co, std=0.05)
        u0 = F.normalize(detgen.normal((co,), 1), dim=0)
        v0 = F.normalize(detgen.normal((ci * k * k,), 2), dim=0)
        for training in (True, False):
            sd = {"c.weight_orig": w.clone().requires_grad_(True), "c.weight_u": u0.clone(), "c.weight_v": v0.clone()}
            wsn = f1_ref.sn_weight(sd, "c", training)
            g = detgen.normal((co, ci, k, k), 3)
            (wsn * g).sum().backward()
            ud, vd = u0.to(DEV), v0.to(DEV)
            wd, sig = ops.spectral_norm_fwd(w.to(DEV), ud, vd, training)
            assert rel(wd, wsn) < 2e-5
            assert rel(ud, sd["c.weight_u"]) < 2e-5 and rel(vd, sd["c.weight_v"]) < 2e-5
            gw = ops.spectral_norm_bwd(g.to(DEV), wd, ud, vd, sig)
            assert rel(gw, sd["c.weight_orig"].grad) < 5e-5
    w = detgen.uniform((3, 3, 5, 5), 9) + 0.5
    ref = f1_ref.bayar_constrain_(w.clone())
    assert rel(ops.bayar_constrain_(w.to(DEV).contiguous()), ref) < 1e-6


# ----------------------------------------------------------------------------- the networks against the reference-generated fixture
def _nets():
    from video_watermarking_forgery_detection_amd.models.networks import Discriminator
    from video_watermarking_forgery_detection_amd.models.conditional_jpeg_generator import FBCNN, QF_predictor
    return Discriminator, FBCNN, QF_predictor


def _check_param_grads(g, key, net, tol, stride):
    n = 0
    params = dict(net.named_parameters())
    for k in [f[len(key) + 3:] for f in g.files if f.startswith(key + "/g/")]:
        got = params[k].grad
        assert got is not None, k
        ref_norm = float(g[f"{key}/gnorm/{k}"])
        assert abs(got.norm().item() - ref_norm) <= tol * ref_norm + 1e-6, (k, got.norm().item(), ref_norm)
        d = np.abs(detgen.subsample(got.cpu(), stride).numpy() - g[f"{key}/g/{k}"]).max()
        assert d <= tol * max(np.abs(g[f"{key}/g/{k}"]).max(), ref_norm / max(got.numel(), 1) ** 0.5) + 1e-7, (k, d)
        n += 1
    for k in [f[len(key) + 8:] for f in g.files if f.startswith(key + "/nograd/")]:
        assert params[k].grad is None, k
    return n


def test_discriminator_against_reference_fixture(golden):
    Discriminator, _, _ = _nets()
    g = golden("f1")
    net = detgen.fill_f1(Discriminator(in_channels=3, use_sigmoid=True)).to(DEV).train()
    x = detgen.uniform((2, 3, 64, 64), 9100).to(DEV).requires_grad_(True)
    y = net(x)
    assert tuple(y.shape) == (2, 1, 2, 2)
    (y * detgen.normal(tuple(y.shape), 9101).to(DEV)).sum().backward()
    assert rel(y, g["disc/y"]) < 1e-4
    assert rel(x.grad, g["disc/gx"]) < 1e-3
    assert _check_param_grads(g, "disc", net, 2e-3, 997) == 11
    sd = net.state_dict()
    for k in [f[len("disc/after/"):] for f in g.files if f.startswith("disc/after/")]:
        assert rel(detgen.subsample(sd[k].cpu(), 7), g["disc/after/" + k]) < 1e-4, k
    net.eval()
    with torch.no_grad():
        assert rel(net(x), g["disc/y_eval"]) < 1e-4


def test_fbcnn_against_reference_fixture(golden):
    _, FBCNN, _ = _nets()
    g = golden("f1")
    net = detgen.fill_f1(FBCNN(nc=[16, 32, 48, 64], nb=2)).to(DEV).train()
    x = detgen.uniform((2, 3, 36, 44), 9200).to(DEV).requires_grad_(True)
    qf = detgen.uniform((2, 1), 9201).to(DEV).requires_grad_(True)
    y, feats = net(x, qf)
    loss = (y * detgen.normal(tuple(y.shape), 9202).to(DEV)).sum()
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g[f"fbcnn/feat{i}_shape"])
        loss = loss + 0.1 * (f * detgen.normal(tuple(f.shape), 9210 + i).to(DEV)).sum()
        assert rel(detgen.subsample(f.cpu(), 13), g[f"fbcnn/feat{i}"]) < 1e-4, i
    loss.backward()
    assert tuple(y.shape) == (2, 3, 36, 44)
    assert rel(y, g["fbcnn/y"]) < 1e-4
    assert rel(x.grad, g["fbcnn/gx"]) < 1e-3
    assert rel(qf.grad, g["fbcnn/gqf"]) < 1e-3
    assert _check_param_grads(g, "fbcnn", net, 2e-3, 97) > 60


def test_qf_predictor_against_reference_fixture(golden):
    _, _, QF_predictor = _nets()
    g = golden("f1")
    net = detgen.fill_f1(QF_predictor(nc=[16, 32, 48, 64], nb=2, classes=5)).to(DEV).train()
    x = detgen.uniform((2, 3, 32, 32), 9300).to(DEV).requires_grad_(True)
    bayar, qf = net(x)
    ((qf * detgen.normal(tuple(qf.shape), 9301).to(DEV)).sum() + 0.05 * (bayar * detgen.normal(tuple(bayar.shape), 9302).to(DEV)).sum()).backward()
    assert rel(bayar, g["qfp/bayar"]) < 1e-4
    assert rel(qf, g["qfp/qf"]) < 1e-4
    assert rel(net.BayarConv2D.weight, g["qfp/bayar_weight_after"]) < 1e-5
    assert rel(x.grad, g["qfp/gx"]) < 1e-3
    assert _check_param_grads(g, "qfp", net, 2e-3, 97) > 40


def test_qf_predictor_crop_pred_against_reference_fixture(golden):
    """QF_predictor(crop_pred=True) (conditional_jpeg_generator.py:772-784, :817-821): the 192 -> 1 projection resized to 512 x 512 by the
    attack set's bicubic kernels, the ResBlock-free head, and every gradient, against tests/golden/f12x.npz (reference-generated)"""
    _, _, QF_predictor = _nets()
    g = golden("f12x")
    net = detgen.fill_f1(QF_predictor(nc=[16, 32, 48, 64], nb=2, classes=5, crop_pred=True)).to(DEV).train()
    x = detgen.uniform((2, 3, 32, 32), 9800).to(DEV).requires_grad_(True)
    img, qf = net(x)
    wimg = detgen.normal((2, 1, 64, 64), 9802).repeat_interleave(8, 2).repeat_interleave(8, 3).to(DEV)
    ((qf * detgen.normal(tuple(qf.shape), 9801).to(DEV)).sum() + 0.05 * (img * wimg).sum()).backward()
    assert tuple(img.shape) == (2, 1, 512, 512)
    assert rel(img[:, :, ::7, ::5], g["qfpc/img_sub"]) < 1e-4
    assert abs(float(img.double().abs().sum()) - float(g["qfpc/img_abs"])) < 1e-4 * float(g["qfpc/img_abs"])
    assert rel(qf, g["qfpc/qf"]) < 1e-4
    assert rel(x.grad, g["qfpc/gx"]) < 1e-3
    assert _check_param_grads(g, "qfpc", net, 2e-3, 97) > 30


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_networks_in_16_bit_against_the_oracle(dtype):
    Discriminator, FBCNN, QF_predictor = _nets()
    tol = 6e-2 if dtype == torch.bfloat16 else 1.5e-2
    # Discriminator at 128x128, batch 3
    net = detgen.fill_f1(Discriminator(in_channels=3, dtype=dtype)).to(DEV).train()
    sd = f1_ref.params({k: v.cpu() for k, v in net.state_dict().items()})
    x = detgen.uniform((3, 3, 128, 128), 1)
    y = net(x.to(DEV))
    yr = f1_ref.discriminator(sd, x, training=True)
    assert tuple(y.shape) == (3, 1, 4, 4) and rel(y, yr) < tol
    # FBCNN at 64x72 with the default widths, nb=1
    net = detgen.fill_f1(FBCNN(nb=1, dtype=dtype)).to(DEV).train()
    sd = f1_ref.params({k: v.cpu() for k, v in net.state_dict().items()})
    x, qf = detgen.uniform((2, 3, 64, 72), 2), detgen.uniform((2, 1), 3)
    xd = x.to(DEV).requires_grad_(True)
    y, feats = net(xd, qf.to(DEV))
    xr = x.clone().requires_grad_(True)
    yr, fr = f1_ref.fbcnn(sd, xr, qf, nb=1)
    assert rel(y, yr) < tol
    for a, b in zip(feats, fr):
        assert rel(a, b) < tol
    gy = detgen.normal(tuple(yr.shape), 4)
    (y * gy.to(DEV)).sum().backward()
    (yr * gy).sum().backward()
    gtol = 3 * tol if dtype == torch.bfloat16 else 5 * tol      # flips of ReLU masks dominate, so the error goes with sqrt(eps): measured 0.119 (bf16) / 0.044 (f16) with forward errors 1.4e-2 / 1.6e-3
    assert rel2(xd.grad, xr.grad) < gtol, rel2(xd.grad, xr.grad)
    p, pr = dict(net.named_parameters()), sd
    for k in ("m_head.weight", "m_up3.0.weight", "m_tail.bias", "to_gamma_2.0.weight", "qf_embed.0.weight"):
        assert rel2(p[k].grad, pr[k].grad) < gtol, (k, rel2(p[k].grad, pr[k].grad))
    # QF_predictor at 64x64
    net = detgen.fill_f1(QF_predictor(nb=1, classes=4, dtype=dtype)).to(DEV).train()
    sd = f1_ref.params({k: v.cpu() for k, v in net.state_dict().items()})
    x = detgen.uniform((2, 3, 64, 64), 5)
    bayar, qf = net(x.to(DEV))
    br, qr = f1_ref.qf_predictor(sd, x, nb=1)
    assert rel(bayar, br) < tol and rel(qf, qr) < 2 * tol


def test_surface_errors_and_state_dict_round_trip():
    from video_watermarking_forgery_detection_amd import glayers as G
    Discriminator, FBCNN, QF_predictor = _nets()
    net = Discriminator(in_channels=3).to(DEV)
    with pytest.raises(RuntimeError, match="GPU only"):
        net(torch.zeros(1, 3, 32, 32))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 4, 32, 32, device=DEV))
    with pytest.raises(NotImplementedError):
        FBCNN(downsample_mode="avgpool")
    with pytest.raises(ValueError):
        G.Conv2d(8, 8, 3, 1, 1).to(DEV)(torch.zeros(1, 4, 4, 32, device=DEV))
    src = detgen.fill_f1(QF_predictor(nb=1)).state_dict()
    dst = QF_predictor(nb=1)
    missing, unexpected = dst.load_state_dict({k: v.clone() for k, v in src.items()}, strict=True)
    assert not missing and not unexpected
    # the optimiser sees exactly the reference's parameters (spectral norm: weight_orig, never the derived weight)
    names = [n for n, _ in Discriminator(in_channels=3).named_parameters()]
    assert "init_conv.0.weight_orig" in names and "conv5.0.weight" in names and not any(n.endswith(".0.weight") and "conv5" not in n for n in names)
