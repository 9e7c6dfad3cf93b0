"""GPU: the f16 compute path (BASELINE configs[4] names fp16; the reference runs under torch.cuda.amp.autocast() with a GradScaler,
models/IRNcrop_model.py:143,340,407-416).  Same MFMA kernels as bf16 (compiled a second time for f16: v_mfma_f32_16x16x32_f16 /
32x32x16_f16, f32 accumulate), the loss scale kept on the device.

  * the device-side scaler (wm_amp_*) against torch.amp.GradScaler + torch.optim.Adam on the CPU, step for step, including skipped
    steps (injected inf), back-off and growth;
  * per-kernel: the f16 conv / weight-gradient kernels against torch's CPU conv on f16-rounded operands;
  * the full HiDDeN-order step in f16 against the f32 CPU oracle at the f16 bound (10-bit mantissa: 8x tighter than bf16);
  * the C5 model surface (clip + UNet head) in f16: runs, finite, scale adapts, tracks the bf16 path.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import detgen
from oracle import hidden_ref, jpeg_ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_device_grad_scaler_matches_torch_amp():
    from video_watermarking_forgery_detection_amd import ops
    n, steps, bad = 4096, 14, (2, 5, 6)
    torch.manual_seed(3)
    p0 = torch.randn(n)
    cs = [torch.randn(n) * 10 ** float(torch.randint(-4, 1, (1,))) for _ in range(steps)]
    # ---- reference: torch.amp.GradScaler("cpu") + torch.optim.Adam
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    sc = torch.amp.GradScaler("cpu", init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    ref = []
    for t in range(steps):
        opt.zero_grad()
        c = cs[t].clone()
        if t in bad:
            c[7] = float("inf")
        sc.scale((pr * c).sum()).backward()
        sc.step(opt)
        sc.update()
        ref.append((sc.get_scale(), pr.detach().clone()))
    # ---- device: the gradient buffer holds scale * c (what a scaled backward leaves), then found_inf / adam_step_amp / update
    dev = torch.device("cuda", 0)
    amp = ops.AmpState(dev, init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    k = amp.slot()
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for t in range(steps):
        c = cs[t].clone()
        if t in bad:
            c[7] = float("inf")
        g = c.to(dev) * amp.scale          # device scalar: no host read of the scale
        amp.found_inf(k, [ops.sumsq(g)])
        ops.adam_step_amp(p, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.0, amp, k)
        amp.update()
        assert amp.get_scale() == ref[t][0], (t, amp.get_scale(), ref[t][0])
        torch.testing.assert_close(p.cpu(), ref[t][1], rtol=2e-6, atol=2e-7)
    assert amp.step_count(k) == steps - len(bad)                      # skipped steps do not advance Adam's step
    assert {r[0] for r in ref} >= {1024.0, 512.0, 256.0} and ref[-1][0] > min(r[0] for r in ref)   # back-off and growth both happened


@pytest.mark.parametrize("case", [(2, 40, 36, 64, 64), (1, 32, 48, 16, 64), (2, 24, 24, 128, 64)])
def test_f16_conv_and_wgrad_kernels(case):
    """the f16 twins of the wave-specialised / streamed conv and weight-gradient kernels against torch's CPU conv on f16-rounded operands"""
    from video_watermarking_forgery_detection_amd import ops
    B, H, W, Cin, Cout = case
    dt = torch.float16
    x = detgen.normal((B, Cin, H, W), 71).to(dt).float()
    w = detgen.normal((Cout, Cin, 3, 3), 72, std=(2.0 / (9 * Cin)) ** 0.5).to(dt).float()
    sc = detgen.normal((Cin,), 74, mean=1.0, std=0.2)
    sh = detgen.normal((Cin,), 75, std=0.3)
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(a, wr, None, padding=1)
    dy = detgen.normal((B, Cout, H, W), 76).to(dt).float()
    ref.backward(dy)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dt).cuda()   # noqa: E731
    wp = ops.pack_w3x3(w.cuda(), Cout, Cin, dt)
    y, st = ops.conv3x3_fwd(nhwc(x), wp, None, sc.cuda(), sh.cuda(), True)
    assert y.dtype == dt
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref.detach(), rtol=3e-3, atol=3e-3)      # f16 output rounding: 2^-11 relative
    s = st.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.detach().sum((0, 2, 3)), rtol=1e-3, atol=2e-2)
    dw = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad(nhwc(x), Cin, sc.cuda(), sh.cuda(), nhwc(dy), dw, False)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=2e-3, atol=2e-3 * wr.grad.abs().max().item())
    wpt = ops.pack_w3x3(w.cuda(), Cout, Cin, dt, transpose=True)
    if Cin in (64, 32):
        gx, _ = ops.conv3x3_fwd(nhwc(dy), wpt, None, None, None, want_stats=False)
        torch.testing.assert_close(gx.float().cpu().permute(0, 3, 1, 2), a.grad, rtol=3e-3, atol=3e-3 * a.grad.abs().max().item())


def _hidden(size, dtype, amp, noise):
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    h = Hidden(HiDDenConfiguration(H=size, W=size), torch.device("cuda"), noise, None, compute_dtype=dtype, amp=amp)
    for m in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
        detgen.fill_module(m)
    return h


def test_f16_step_vs_oracle():
    from video_watermarking_forgery_detection_amd import ops
    from video_watermarking_forgery_detection_amd.noise_layers import JpegSS
    S, B = 64, 8
    images = detgen.uniform((B, 3, S, S), 9500)
    messages = detgen.bits((B, 30), 9501)
    ref = hidden_ref.HiddenRef(hidden_ref.HiDDenConfiguration(H=S, W=S), lambda x: jpeg_ref.jpeg_layer(x, 50, "ss"))
    for m in (ref.encoder, ref.decoder, ref.discriminator):
        detgen.fill_module(m)
    rl, (renc, _, rdec), rgrads = ref.train_on_batch(images, messages)
    amp = ops.AmpState(torch.device("cuda", 0))
    h = _hidden(S, torch.float16, amp, JpegSS(50))
    losses, (e, _, d) = h.train_on_batch([images, messages])
    rep = dict(enc=rel(e, renc), dec=rel(d, rdec), loss={k.strip(): abs(losses[k] - rl[k]) / max(1.0, abs(rl[k])) for k in rl})
    wd = []
    for mine, r in ((h.encoder_decoder.encoder, ref.encoder), (h.encoder_decoder.decoder, ref.decoder), (h.discriminator, ref.discriminator)):
        for (n, p), (_, q) in zip(mine.state_dict().items(), r.state_dict().items()):
            if not n.endswith("num_batches_tracked"):
                wd.append((p.float().cpu() - q).abs().flatten())
    wd = torch.cat(wd)
    rep["w_mean"], rep["w_max"], rep["scale"] = float(wd.mean()), float(wd.max()), amp.get_scale()
    print("f16", rep)
    # f16 activations: 2^-11 relative rounding per layer (bf16: 2^-9) -> a quarter of the bf16 bounds of test_gpu_configs
    assert rep["enc"] < 5e-3 and rep["dec"] < 1.5e-2, rep
    assert all(v < 5e-3 for v in rep["loss"].values()), rep
    assert rep["w_mean"] < 3e-4 and rep["w_max"] <= 2.1e-3, rep       # the optimiser stepped on correctly un-scaled gradients
    assert rep["scale"] in (65536.0, 32768.0, 16384.0), rep            # no runaway back-off: the scaled gradients fit f16
    assert amp.step_count(h.optimizer_enc_dec.amp_slot) + amp.step_count(h.optimizer_discrim.amp_slot) >= 1


def test_f16_c5_model_surface_tracks_bf16(tmp_path):
    """16-frame clip + UNet head through feed_data / optimize_parameters in f16 (train.compute_dtype: f16 turns the scaler on):
    finite, parameters move, and the logged losses stay within the 16-bit bounds of the bf16 run on the same seeds"""
    from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
    from video_watermarking_forgery_detection_amd.options.options import dict_to_nonedict
    S, T = 64, 16
    out = {}
    for name in ("f16", "bf16"):
        opt = dict_to_nonedict({"gpu_ids": [0], "dist": False, "is_train": True, "datasets": {"train": {"GT_size": S, "batch_size": 1}},
                                "train": {"compute_dtype": name, "attacks": ["JpegSS70", "GaussianBlur", "Resize"], "lr_G": 1e-3, "localizer": True,
                                          "gradient_clipping": 1.0, "save_interval": 3000},
                                "path": {"models": str(tmp_path / name)}})
        m = IRNrhiModel(opt)
        assert (m.amp is not None) == (name == "f16")
        for net in (m.netG.encoder, m.netG.decoder, m.discriminator, m.localizer):
            detgen.fill_module(net)
        w0 = m.localizer.flat_params.clone()
        logs_all = []
        for step in range(1, 7):
            clip = detgen.uniform((1, 3, T, S, S), 9600 + step)
            mask = torch.zeros(1, 1, T, S, S)
            mask[..., 16:48, 8:40] = 1.0
            m.feed_data({"GT": clip, "mask": mask, "messages": detgen.bits((T, 30), 9700 + step)})
            logs, _ = m.optimize_parameters(step, None)
            if logs:
                d = dict(logs)
                assert all(np.isfinite(v) for v in d.values() if isinstance(v, float)), d
                logs_all.append(d)
        assert not torch.equal(w0, m.localizer.flat_params) and torch.isfinite(m.localizer.flat_params).all()
        out[name] = logs_all
        if name == "f16":
            assert m.amp.get_scale() >= 1024.0 and m.amp.step_count(m.optimizer_localizer.amp_slot) >= 3
    # the first trained step agrees at the 16-bit bounds; after that two differently-rounded trainings of a random-init GAN drift apart
    # step by step (the parameters differ after Adam's sign-like first updates): only a loose bound further on
    for i, (a, b) in enumerate(zip(out["f16"], out["bf16"])):
        for k in ("encoder_mse", "dec_mse", "lB"):
            assert abs(a[k] - b[k]) < (5e-2 if i == 0 else 2.5e-1) * max(1.0, abs(b[k])), (i, k, a[k], b[k])


def test_f16_resume_continues_the_run():
    """save -> load -> step under the device-side scaler: a resumed f16 run must step exactly like the run it was saved from.  Adam's
    bias-correction t lives in the scaler's device state (adam_amp_kernel reads state[12 + slot]); round 2's load_state_dict restored
    the moments and the host count only, so the first resumed update ran with t = 1 on warm moments (bc1 = 0.1, bc2 = 0.001)."""
    from video_watermarking_forgery_detection_amd import ops
    from video_watermarking_forgery_detection_amd.noise_layers import JpegSS
    S, B, K = 32, 4, 3
    images = detgen.uniform((B, 3, S, S), 9600)
    messages = detgen.bits((B, 30), 9601)
    amp_a = ops.AmpState(torch.device("cuda", 0))
    a = _hidden(S, torch.float16, amp_a, JpegSS(50))
    for _ in range(K):
        a.train_on_batch([images, messages])
    nets = lambda h: (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator)   # noqa: E731
    saved = dict(nets=[{k: v.clone() for k, v in m.state_dict().items()} for m in nets(a)], opt_ed=a.optimizer_enc_dec.state_dict(),
                 opt_d=a.optimizer_discrim.state_dict(), amp=amp_a.state_dict())
    assert float(saved["opt_ed"]["state"][0]["step"]) == amp_a.step_count(a.optimizer_enc_dec.amp_slot) >= 1
    amp_b = ops.AmpState(torch.device("cuda", 0))
    b = _hidden(S, torch.float16, amp_b, JpegSS(50))
    for m, sd in zip(nets(b), saved["nets"]):
        m.load_state_dict(sd)
    b.optimizer_enc_dec.load_state_dict(saved["opt_ed"])
    b.optimizer_discrim.load_state_dict(saved["opt_d"])
    amp_b.load_state_dict(saved["amp"])
    assert amp_b.step_count(b.optimizer_enc_dec.amp_slot) == amp_a.step_count(a.optimizer_enc_dec.amp_slot)
    assert amp_b.get_scale() == amp_a.get_scale() and amp_b.state_dict() == amp_a.state_dict()
    la, _ = a.train_on_batch([images, messages])
    lb, _ = b.train_on_batch([images, messages])
    assert dict(la) == dict(lb)
    for ma, mb in zip(nets(a), nets(b)):
        assert torch.equal(ma.flat_params, mb.flat_params)          # the resumed step IS the continued step
    assert b.optimizer_enc_dec.state_dict()["state"][0]["step"] == a.optimizer_enc_dec.state_dict()["state"][0]["step"]
