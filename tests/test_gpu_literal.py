"""GPU: the reference's literal IRNrhi step (models/IRNrhi_model.py:425-560: QF_predictor + FBCNN + Discriminator, SmoothL1 / BCE /
cross-entropy losses, AdamW with gradient clipping) on the HIP layer toolkit.

  * the loss / clamp / PSNR kernels (csrc/losses.hip) against torch's own modules, values and gradients;
  * FlatAdamW (one flat buffer, the library's Adam kernel with decoupled decay) against torch.optim.AdamW, with clipping;
  * three consecutive steps of models/IRNrhi_literal.IRNrhiLiteralModel against the CPU composition of the pinned oracle networks
    (oracle/irnrhi_literal_ref.py): every logged scalar, the simulated images and the updated parameters.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

import detgen
from oracle import irnrhi_literal_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_loss_kernels_against_torch():
    from video_watermarking_forgery_detection_amd import glayers as G, ops
    a = detgen.normal((3, 3, 17, 19), 1, std=1.5)
    b = detgen.normal((3, 3, 17, 19), 2)
    ar = a.clone().requires_grad_(True)
    ref = nn.SmoothL1Loss()(ar, b)
    (ref * 0.7).backward()
    ad = a.to(DEV).requires_grad_(True)
    out = G.smooth_l1_loss(ad, b.to(DEV))
    (out * 0.7).backward()
    assert abs(out.item() - ref.item()) < 1e-6 * max(1, abs(ref.item())) and rel(ad.grad, ar.grad) < 1e-5

    p = torch.sigmoid(detgen.normal((7, 1, 3, 3), 3, std=3.0))
    p[0, 0, 0, 0], p[1, 0, 0, 0] = 0.0, 1.0                      # the clamped-log corners
    for target in (1.0, 0.0):
        pr = p.clone().requires_grad_(True)
        ref = nn.BCELoss()(pr, torch.full_like(pr, target))
        ref.backward()
        pd = p.to(DEV).requires_grad_(True)
        out = G.bce_loss(pd, target)
        out.backward()
        assert abs(out.item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
        assert rel(pd.grad, pr.grad) < 1e-5

    z = detgen.normal((24, 6), 4, std=2.0)
    y = torch.arange(6).repeat_interleave(4)
    zr = z.clone().requires_grad_(True)
    ref = nn.CrossEntropyLoss()(zr, y)
    (ref * 0.01).backward()
    zd = z.to(DEV).requires_grad_(True)
    out = G.cross_entropy_loss(zd, y.to(DEV))
    (out * 0.01).backward()
    assert abs(out.item() - ref.item()) < 1e-6 * max(1, abs(ref.item())) and rel(zd.grad, zr.grad) < 1e-5

    x = detgen.normal((2, 3, 8, 8), 5)
    x.view(-1)[:4] = torch.tensor([0.0, 1.0, -0.0, 1.0000001])
    xr = x.clone().requires_grad_(True)
    g = detgen.normal((2, 3, 8, 8), 6)
    (torch.clamp(xr, 0, 1) * g).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    yd = G.clamp01(xd)
    (yd * g.to(DEV)).sum().backward()
    assert torch.equal(yd.cpu(), torch.clamp(x, 0, 1)) and torch.equal(xd.grad.cpu(), xr.grad)

    i1, i2 = detgen.uniform((2, 3, 16, 16), 7), detgen.uniform((2, 3, 16, 16), 8)
    mse = torch.mean(((i1 * 255.0).int().float() - (i2 * 255.0).int().float()) ** 2)
    ref = 20 * np.log10(255.0) - 10 * np.log10(mse.item())
    assert abs(ops.psnr255(i1.to(DEV), i2.to(DEV)).item() - ref) < 1e-3
    assert ops.psnr255(i1.to(DEV), i1.to(DEV)).item() == 0.0          # metrics.py:41-42


def test_flat_adamw_against_torch():
    from video_watermarking_forgery_detection_amd import glayers as G
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Linear(13, 7), nn.Linear(7, 3))
    net = nn.Sequential(nn.Linear(13, 7), nn.Linear(7, 3))
    net.load_state_dict(ref.state_dict())
    net.to(DEV)
    o_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.8, 0.95), weight_decay=0.05)
    o = G.FlatAdamW(net, 1e-2, (0.8, 0.95), weight_decay=0.05)
    for it in range(4):
        x = detgen.normal((5, 13), 10 + it)
        o_ref.zero_grad(); o.zero_grad()
        (ref(x) ** 2).sum().backward()
        (net(x.to(DEV)) ** 2).sum().backward()
        nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        o.clip_grad_norm_(1.0)
        o_ref.step(); o.step()
        for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
            assert rel(a, b) < 1e-5, (it, k)
    with pytest.raises(RuntimeError):
        G.FlatAdamW(nn.Linear(2, 2), 1e-3)           # parameters on the CPU


def test_literal_step_against_the_cpu_composition():
    from video_watermarking_forgery_detection_amd.models.IRNrhi_literal import IRNrhiLiteralModel
    torch.manual_seed(1)
    opt = {"gpu_ids": [0], "is_train": True, "dist": False, "network": {"nc": [16, 32, 48, 64], "nb": 1},
           "train": {"lr_D": 2e-4, "beta1": 0.9, "beta2": 0.999, "weight_decay_G": 0.01, "gradient_clipping": 1.0, "compute_dtype": "f32"}}
    model = IRNrhiLiteralModel(opt)
    with torch.no_grad():
        model.localizer.BayarConv2D.weight.copy_(detgen.uniform((3, 3, 5, 5), 77).to(DEV) + 0.5)

    def cpu_sd(net):
        return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}

    ref = irnrhi_literal_ref.LiteralRef(cpu_sd(model.generator), cpu_sd(model.localizer), cpu_sd(model.discriminator), nb=1, lr=2e-4,
                                        weight_decay=0.01, clip=1.0)
    bs = 2
    for it in range(3):
        base = detgen.uniform((bs, 3, 32, 32), 500 + it)
        imgs = [torch.clamp(base + 0.02 * q * detgen.normal((bs, 3, 32, 32), 600 + 10 * it + q), -0.1, 1.1) for q in range(6)]
        model.feed_data((imgs, None))
        logs, _ = model.optimize_parameters(it)
        want = ref.step(imgs)
        got = dict(logs)
        assert list(got) == ['l_simul_bayar', 'FW_GAN', 'lQF', 'PSSIMU', 'qfsimu']
        got.update(dis_loss=model.last["dis_loss"], l_simul_sum=model.last["l_simul_sum"])
        for k, v in got.items():
            assert abs(v - want[k]) <= 2e-3 * max(abs(want[k]), 1e-2), (it, k, v, want[k])
        assert rel(model.last["simulated"], want["simulated"]) < 5e-3, it
    for net, sd in ((model.generator, ref.g), (model.localizer, ref.l), (model.discriminator, ref.d)):
        for k, v in net.state_dict().items():
            if k.endswith("haar_weights"):
                continue
            diff = (v.cpu() - sd[k].detach()).abs()
            # Adam's step is sign-like where a gradient is at round-off level, so single elements may move the other way: each of the
            # 3 iterations moves an element by at most ~lr (2e-4); the bulk must agree far more closely
            assert diff.max().item() <= 2 * 3 * 2e-4 + 1e-5, (k, diff.max().item())
            assert diff.mean().item() <= 2e-5, (k, diff.mean().item())
    # unfed / ragged batch: the reference skips the step
    model.feed_data(([torch.zeros(1, 3, 32, 32)] * 5, None))
    assert model.optimize_parameters(9) == ([], [])
