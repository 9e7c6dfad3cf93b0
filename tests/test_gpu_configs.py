"""GPU: the BASELINE.json configurations C2-C5 against pinned numbers.

  C2  256x256, batch 16, Jpeg(50): the benchmarked bf16 kernels AND the f32 parity path against the CPU oracle, full size
  C3  the HiDDeN-order step under each attack of the cycle (GaussianBlur, Resize 0.7, Crop 0.75, MiddleBlur 3) against the
      fixture generated from the reference's modules (tests/golden/step_c3.npz), incl. discriminator gradients and `noised`
  C4  the per-GPU shard of the 512x512 data-parallel configuration (8 frames) against the CPU oracle
  C5  16-frame clip + UNet tamper-localisation head: the localisation branch against the fixture composed from the
      reference's modules (tests/golden/localise.npz) and, at 16 x 256x256, against the CPU oracle; integer tamper masks bit-exact
north_star tolerance: 1e-3 relative (f32 path) on watermarked / noised / decoded tensors; the bf16 bounds are stated per test.
"""
import numpy as np
import pytest
import torch

import detgen
from oracle import attacks_ref, hidden_ref, jpeg_ref, localise_ref, unet_ref

pytestmark = pytest.mark.gpu

KEYS = ("loss           ", "encoder_mse    ", "dec_mse        ", "bitwise-error  ", "adversarial_bce", "discr_cover_bce",
        "discr_encod_bce")


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def rel_l2(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))


def make_hidden(size, noise, dtype):
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    h = Hidden(HiDDenConfiguration(H=size, W=size), torch.device("cuda"), noise, None, compute_dtype=dtype)
    for m in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
        detgen.fill_module(m)
    return h


def make_ref(size, noise_fn):
    ref = hidden_ref.HiddenRef(hidden_ref.HiDDenConfiguration(H=size, W=size), noise_fn)
    for m in (ref.encoder, ref.decoder, ref.discriminator):
        detgen.fill_module(m)
    return ref


def nets(h):
    return (("E", h.encoder_decoder.encoder), ("Dec", h.encoder_decoder.decoder), ("D", h.discriminator))


def check_against_oracle(h32, hbf, ref, images, messages, tag):
    """one step of the f32 path and of the bf16 production path against one oracle step on the same seeds"""
    rl, (renc, rnoised, rdec), rgrads = ref.train_on_batch(images, messages)
    rw = {k: {n: p.detach().clone() for n, p in m.state_dict().items()} for k, m in
          (("E", ref.encoder), ("Dec", ref.decoder), ("D", ref.discriminator))}
    report = {}
    for name, h in (("f32", h32), ("bf16", hbf)):
        losses, (e, nz, d) = h.train_on_batch([images, messages])
        report[name] = dict(enc=rel(e, renc), dec=rel(d, rdec),
                            noised_flip=float(((nz.cpu() - rnoised.detach()).abs() > 1e-3).float().mean()),
                            loss={k.strip(): abs(losses[k] - rl[k]) / max(1.0, abs(rl[k])) for k in rl})
        # post-Adam parameters: one Adam step moves every weight by <= lr = 1e-3
        wd = []
        for k, m in nets(h):
            for n, p in m.state_dict().items():
                if n.endswith("num_batches_tracked"):
                    assert int(p) == int(rw[k][n]), (name, k, n)
                    continue
                wd.append((p.float().cpu() - rw[k][n]).abs().flatten())
        wd = torch.cat(wd)
        report[name]["w_mean"], report[name]["w_max"] = float(wd.mean()), float(wd.max())
        gl2 = {}
        for k, m in nets(h)[:2]:   # (the discriminator's .grad also holds the generator pass by now, like the reference's: see C3)
            for n, p in m.named_parameters():
                if n.endswith("layers.0.bias"):      # exactly zero in front of a training-mode BatchNorm
                    continue
                gl2[f"{k}.{n}"] = rel_l2(p.grad, rgrads[k][n])
        report[name]["grad_l2_max"] = max(gl2.values())
        report[name]["grad_l2_argmax"] = max(gl2, key=gl2.get)
        report[name]["grad_l2_median"] = float(np.median(list(gl2.values())))
    print(tag, report)
    f, b = report["f32"], report["bf16"]
    # ---- f32 parity path: north_star's 1e-3 on the watermarked / decoded tensors and the losses
    assert f["enc"] < 1e-3 and f["dec"] < 1e-3, f
    assert all(v < 1e-3 for v in f["loss"].values()), f
    assert f["noised_flip"] < 2e-2, f               # Jpeg(50): a coefficient within f32 round-off of .5 may round the other way
    assert f["w_mean"] < 3e-4 and f["w_max"] <= 2.1e-3, f
    assert f["grad_l2_median"] < 1e-2 and f["grad_l2_max"] < 5e-2, f
    # ---- bf16 production path (what bench.py times): activations are rounded to bf16 (2^-9 relative) once per layer, f32
    # accumulation and statistics; through the 5-layer encoder and the 8-layer decoder that is a few 1e-3 of the output range
    assert b["enc"] < 2e-2 and b["dec"] < 5e-2, b
    assert all(v < 2e-2 for v in b["loss"].values()), b
    assert b["w_mean"] < 6e-4 and b["w_max"] <= 2.1e-3, b
    assert b["grad_l2_median"] < 8e-2, b
    # ... and NO single parameter's gradient may be badly off (round 3's verdict: a median alone would let one through).  Measured maxima of
    # the relative L2 error over all encoder / decoder parameters (round 4, MI355X): C2 (Jpeg50) 0.122, C4 shard (JpegSS50, 512^2) 0.065,
    # full-size C3 under Resize(0.7) 0.109, under GaussianBlur 0.059 (the report names the parameter: grad_l2_argmax).  Bound: 2.5 x the
    # largest of them.  (The f32 path's maximum on the same steps is 1e-3..3e-3: the bf16 figure is rounding noise of gradients that
    # are sums of ~1e6 bf16-rounded terms which largely cancel, not an error of the arithmetic -- tests/test_gpu_bwd_oracle.py
    # bounds that noise per kernel.)
    assert b["grad_l2_max"] < 0.3, b
    return report


def test_c2_full_size_f32_and_bf16_vs_oracle():
    """BASELINE configs[1] as benchmarked: B=16, 256x256, Jpeg(50) -- the bf16 kernel family bench.py runs (wave-specialised conv /
    wgrad, fused BatchNorm-backward modes, the streamed concat layer) end to end against the CPU oracle, beside the f32 path."""
    from video_watermarking_forgery_detection_amd.noise_layers import Jpeg
    S, B = 256, 16
    images = detgen.uniform((B, 3, S, S), 9100)
    messages = detgen.bits((B, 30), 9101)
    ref = make_ref(S, lambda x: jpeg_ref.jpeg_layer(x, 50, "round"))
    check_against_oracle(make_hidden(S, Jpeg(50), torch.float32), make_hidden(S, Jpeg(50), torch.bfloat16), ref, images, messages, "C2")


def test_c4_shard_512_vs_oracle():
    """BASELINE configs[3], one rank's shard: 8 frames of 512x512 (global batch 64 over 8 GPUs), JpegSS(50) so that the attack
    passes gradient to the encoder."""
    from video_watermarking_forgery_detection_amd.noise_layers import JpegSS
    S, B = 512, 8
    images = detgen.uniform((B, 3, S, S), 9200)
    messages = detgen.bits((B, 30), 9201)
    ref = make_ref(S, lambda x: jpeg_ref.jpeg_layer(x, 50, "ss"))
    check_against_oracle(make_hidden(S, JpegSS(50), torch.float32), make_hidden(S, JpegSS(50), torch.bfloat16), ref, images, messages, "C4")


@pytest.mark.parametrize("nname", ["Resize0.7", "GaussianBlur"])
def test_c3_full_size_step_vs_oracle(nname):
    """config C3 at FULL size (B=16, 256x256) for two attacks of the cycle, f32 parity path and bf16 production path against one oracle
    step (the oracle's attack ops are pinned by tests/golden/attacks.npz, its step by step_c3.npz at 32x32): the fixture-size tests
    above cannot see a full-size-only failure of the separable resample backward or the stencil (tile seams, 32-bit offsets)."""
    S, B = 256, 16
    images = detgen.uniform((B, 3, S, S), 9300)
    messages = detgen.bits((B, 30), 9301)
    fn = {"Resize0.7": lambda x: attacks_ref.resize(x, 0.7), "GaussianBlur": lambda x: attacks_ref.gaussian_blur(x)}[nname]
    ref = make_ref(S, fn)
    rep = check_against_oracle(make_hidden(S, _c3_layers()[nname], torch.float32), make_hidden(S, _c3_layers()[nname], torch.bfloat16), ref,
                               images, messages, f"C3/{nname}")
    # a smooth attack has no rounding decisions: `noised` itself must agree, not just a bounded fraction of it
    assert rep["f32"]["noised_flip"] == 0.0, rep["f32"]


class _Fixed:
    """an attack layer with its random arguments pinned (the fixtures use resize_ratio 0.7 and the 0.75 x 0.75 crop apex)"""

    def __init__(self, layer, **kw):
        self.layer, self.kw = layer, kw
        self.name = getattr(layer, "name", type(layer).__name__)

    def fwd(self, x):
        kw = dict(self.kw)
        if kw.pop("crop75", False):
            H, W = x.shape[2], x.shape[3]
            kw["apex"] = (H // 8, H // 8 + int(0.75 * H), W // 8, W // 8 + int(0.75 * W))
        return self.layer.fwd(x, **kw)

    def bwd(self, c, g):
        return self.layer.bwd(c, g)


def _c3_layers():
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    return {"GaussianBlur": _Fixed(NL.GaussianBlur()), "Resize0.7": _Fixed(NL.Resize(), resize_ratio=0.7),
            "Crop0.75": _Fixed(NL.Crop(), crop75=True), "MiddleBlur3": _Fixed(NL.MiddleBlur(3))}


@pytest.mark.parametrize("nname", ["GaussianBlur", "Resize0.7", "Crop0.75", "MiddleBlur3"])
def test_c3_step_golden(golden, nname):
    """config C3: the full HiDDeN-order step (with discriminator) under each stencil / resample attack, f32 path, against the
    fixture generated from the reference's modules -- losses of two iterations, encoded / noised / decoded, the gradients of all
    three networks after the first iteration and the parameters after two Adam steps.  (MiddleBlur3: kornia is absent from the
    image, the fixture holds the build's own definition of the median: parity unpinned for that op.)"""
    g = golden("step_c3")
    h = make_hidden(32, _c3_layers()[nname], torch.float32)
    enc, dec, dis = h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator
    images = detgen.uniform((4, 3, 32, 32), 2000)
    messages = detgen.bits((4, 30), 2001)
    pre = f"step_{nname}"
    for it in range(2):
        if it == 0:
            # the discriminator's gradients as the reference has them at opt_d.step(): sampled through a clip hook (before the
            # generator pass adds to them)
            seen = {}
            losses, (e, nz, d) = h.train_on_batch([images, messages], clip=lambda flats: seen.setdefault(len(seen), [f.clone() for f in flats]))
        else:
            losses, (e, nz, d) = h.train_on_batch([images, messages])
        got = np.array([losses[k] for k in KEYS])
        np.testing.assert_allclose(got, g[f"{pre}/losses_it{it}"], rtol=5e-3 if it else 1e-3, atol=1e-4)
        if it == 0:
            assert rel(e, g[f"{pre}/encoded"]) < 1e-3
            assert rel(nz, g[f"{pre}/noised"]) < 1e-3
            assert rel(d, g[f"{pre}/decoded"]) < 1e-3
            gD_flat = seen[0][0]
            off = 0
            for n, p in dis.named_parameters():
                k = p.numel()
                got_g = detgen.subsample(gD_flat[off:off + k], 31).cpu().numpy()
                off += k
                if n.endswith("layers.0.bias"):
                    assert np.abs(got_g).max() < 1e-5, n
                    continue
                ref_g = g[f"{pre}/gD/{n}"]
                assert np.abs(got_g - ref_g).max() < 5e-2 * np.abs(ref_g).max() + 1e-7, ("gD", n)
                assert np.linalg.norm(got_g - ref_g) < 3e-2 * np.linalg.norm(ref_g) + 1e-7, ("gD", n)
            for tag, mod in (("gE", enc), ("gDec", dec)):
                for n, p in mod.named_parameters():
                    if n.endswith("layers.0.bias"):
                        assert p.grad.abs().max().item() < 1e-5, n
                        continue
                    ref_g = g[f"{pre}/{tag}/{n}"]
                    got_g = detgen.subsample(p.grad, 31).cpu().numpy()
                    assert np.abs(got_g - ref_g).max() < 5e-2 * np.abs(ref_g).max() + 1e-7, (tag, n)
                    # (a 64-element BatchNorm vector leaves 3 samples: the L2 bound needs a population)
                    assert np.linalg.norm(got_g - ref_g) < (3e-2 if ref_g.size >= 16 else 1e-1) * np.linalg.norm(ref_g) + 1e-7, (tag, n)
    for tag, m in (("wE", enc), ("wDec", dec), ("wD", dis)):
        diffs = []
        for n, p in m.state_dict().items():
            ref_w = g[f"{pre}/{tag}/{n}"]
            got_w = detgen.subsample(p.float(), 31).cpu().numpy()
            if n.endswith("num_batches_tracked"):
                assert np.array_equal(got_w, ref_w), (tag, n)
                continue
            dd = np.abs(got_w - ref_w)
            assert dd.max() <= 4e-3 + 1e-3 * np.abs(ref_w).max(), (tag, n)
            diffs.append(dd)
        dd = np.concatenate(diffs)
        assert dd.mean() < 3e-4 and (dd > 1e-3).mean() < 0.1, (tag, dd.mean())


def test_golden_step_discriminator_grads_and_noised(golden):
    """the Jpeg / Identity step fixtures of round 1 also hold gD/* and `noised`: compare them too (Jpeg50's hard rounding may
    flip a coefficient that sits within f32 round-off of .5: bounded fraction of affected pixels)"""
    from video_watermarking_forgery_detection_amd.noise_layers import Jpeg, JpegSS, JpegMask, Identity
    g = golden("step")
    for nname, noise in (("JpegSS50", JpegSS(50)), ("Jpeg50", Jpeg(50)), ("JpegMask50", JpegMask(50)), ("Identity", Identity())):
        h = make_hidden(32, noise, torch.float32)
        images = detgen.uniform((4, 3, 32, 32), 2000)
        messages = detgen.bits((4, 30), 2001)
        seen = {}
        _, (e, nz, d) = h.train_on_batch([images, messages], clip=lambda flats: seen.setdefault(len(seen), [f.clone() for f in flats]))
        ref_n = g[f"step_{nname}/noised"]
        if nname == "Jpeg50":
            assert (np.abs(nz.cpu().numpy() - ref_n) > 1e-3).mean() < 2e-2
        else:
            assert rel(nz, ref_n) < 1e-3
        off = 0
        for n, p in h.discriminator.named_parameters():
            k = p.numel()
            got_g = detgen.subsample(seen[0][0][off:off + k], 31).cpu().numpy()
            off += k
            if n.endswith("layers.0.bias"):
                continue
            ref_g = g[f"step_{nname}/gD/{n}"]
            assert np.linalg.norm(got_g - ref_g) < 3e-2 * np.linalg.norm(ref_g) + 1e-7, (nname, n)


# --------------------------------------------------------------------------------------------- C5 / row a18
def make_opt(tmp_path, size, **train):
    from video_watermarking_forgery_detection_amd.options.options import dict_to_nonedict
    t = {"compute_dtype": "f32", "attacks": ["JpegSS70"], "lr_G": 1e-3, "manual_seed": 10, "save_interval": 3000, "localizer": True}
    t.update(train)
    return dict_to_nonedict({"gpu_ids": [0], "dist": False, "is_train": True, "datasets": {"train": {"GT_size": size, "batch_size": 4}},
                             "train": t, "path": {"models": str(tmp_path / "models"), "training_state": str(tmp_path / "state")}})


def make_model(tmp_path, size, **train):
    from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
    m = IRNrhiModel(make_opt(tmp_path, size, **train))
    for net in (m.netG.encoder, m.netG.decoder, m.discriminator, m.localizer):
        detgen.fill_module(net)
    m.keep_outputs = True
    return m


_ATTACK_OPT = {"JpegSS70": "JpegSS70", "Resize0.7": "Resize", "GaussianBlur": "GaussianBlur"}


@pytest.mark.parametrize("case", ["jpegss_clip", "resize", "gauss_hipsnr"])
def test_localise_branch_golden(golden, tmp_path, case):
    """row a18 behind feed_data / optimize_parameters, f32 path, against the fixture composed from the reference's modules:
    clamp-STE -> Quantization -> splice -> attack -> clamp-STE -> Quantization -> UNet -> BCEWithLogits(sigmoid mask), the
    PSNR-gated fidelity weight (gauss_hipsnr crosses 33 dB: 0.8 in the first iteration, 1.0 in the second), joint gradient
    clipping over encoder+decoder, and the integer tamper mask."""
    g = golden("localise")
    key = f"loc_{case}"
    clip = float(g[key + "/clip"]) or None
    gain = float(g[key + "/enc_gain"])
    m = make_model(tmp_path, 32, attacks=[_ATTACK_OPT[str(g[key + "/attack"])]], gradient_clipping=clip)
    B, size = 4, 32
    images = detgen.uniform((B, 3, size, size), 2100)
    if gain != 1.0:
        with torch.no_grad():
            m.netG.encoder.final_layer.weight.mul_(gain)
            m.netG.encoder.final_layer.bias.copy_(torch.tensor([0.4, 0.5, 0.6]))
        images = torch.tensor([0.4, 0.5, 0.6]).view(1, 3, 1, 1).expand(B, 3, size, size).contiguous() + 0.01 * (images - 0.5)
    messages = detgen.bits((B, 30), 2101)
    previous = detgen.uniform((B, 3, size, size), 2102).cuda()
    mask = torch.zeros(B, 1, size, size)
    mask[:, :, 8:24, 4:20] = 1.0
    mask[1] = 0.0
    for it in range(2):
        m.previous_images = previous
        m.previous_previous_images = previous
        m.feed_data({"GT": images, "mask": mask, "messages": messages})
        unet_before = _unet_snapshot(m)
        logs, _ = m.optimize_parameters(it + 1, None)
        d = dict(logs)
        ref = g[f"{key}/logs_it{it}"]     # loss, enc, dec, adv, d_cover, d_enc, PF, gate, lB
        got = [d["loss"], d["encoder_mse"], d["dec_mse"], d["adversarial_bce"], d["discr_cover_bce"], d["discr_encod_bce"], d["PF"], d["lB"]]
        np.testing.assert_allclose(got, ref[[0, 1, 2, 3, 4, 5, 6, 8]], rtol=5e-3 if it else 1e-3, atol=1e-4)
        if it == 0:
            o = m.last_outputs
            assert rel(o["encoded"], g[f"{key}/encoded"]) < 1e-3
            assert rel(o["decoded"], g[f"{key}/decoded"]) < 1e-3
            # quantised tensors: equal up to a flipped rounding step where 255*x sits within round-off of .5
            # (`attacked`: a flipped step of `tampered` goes through the attack first -- JpegSS's cubic rounding has slope up to 3 -- so a
            # few of its pixels may land up to three steps away)
            for nm, steps, frac in (("tampered", 1, 1e-3), ("attacked", 3, 5e-3)):
                dq = np.abs(o[nm].cpu().numpy() - g[f"{key}/{nm}"])
                assert dq.max() <= steps / 255 + 1e-6 and (dq > 1e-6).mean() < frac, (nm, float(dq.max()), float((dq > 1e-6).mean()))
            # the predicted mask: 1e-3 against the oracle UNet on the same input; against the fixture (whose input differs by the
            # flipped steps above, which shift every BatchNorm's batch statistics) only a sanity bound
            assert rel(o["pred"], _pred_on_same_input(unet_before, o["attacked"])) < 1e-3
            assert np.abs(o["pred"].cpu().numpy() - g[f"{key}/pred"]).max() < 3e-2
            # the UNet's parameter gradients: against the oracle UNet's autograd on the same input (the fixture's were taken on an
            # input that differs by the flipped steps, and its 997-strided samples of the 64-element BatchNorm vectors are single values)
            ref_u = unet_before
            ref_u.zero_grad()
            pr = ref_u(o["attacked"].detach().cpu().float())
            torch.nn.BCEWithLogitsLoss()(pr, mask).backward()
            if clip:
                torch.nn.utils.clip_grad_norm_(ref_u.parameters(), clip)
            for (n, p), (_, q) in zip(m.localizer.named_parameters(), ref_u.named_parameters()):
                assert rel_l2(p.grad, q.grad) < 3e-2, ("gU", n, rel_l2(p.grad, q.grad))
    # parameters after two Adam steps
    for tag, mod, st in (("wE", m.netG.encoder, 31), ("wDec", m.netG.decoder, 31), ("wD", m.discriminator, 31), ("wU", m.localizer, 997)):
        diffs = []
        for n, p in mod.state_dict().items():
            ref_w = g[f"{key}/{tag}/{n}"]
            got_w = detgen.subsample(p.float(), st).cpu().numpy()
            if n.endswith("num_batches_tracked"):
                assert np.array_equal(got_w, ref_w), (tag, n)
                continue
            dd = np.abs(got_w - ref_w)
            if tag == "wU" and n.endswith(("running_mean", "running_var")):
                # running statistics are not Adam-stepped: they follow the batch statistics, which the flipped quantisation steps of
                # the UNet's input shift a little (the UNet itself is pinned on identical inputs above)
                assert dd.max() <= 2e-2 * max(1.0, np.abs(ref_w).max()), (tag, n)
                continue
            assert dd.max() <= 4e-3 + 1e-3 * np.abs(ref_w).max(), (tag, n)
            diffs.append(dd)
        dd = np.concatenate(diffs)
        # (the UNet's gradients were taken on an input that differs from the fixture's by the flipped quantisation steps; Adam's
        # sign-like first steps turn that into a slightly wider spread of its parameters)
        assert dd.mean() < (6e-4 if tag == "wU" else 3e-4) and (dd > 1e-3).mean() < (0.2 if tag == "wU" else 0.1), (tag, dd.mean())
    _check_integer_mask(m, images)


def _unet_snapshot(m):
    """the oracle UNet holding the localiser's CURRENT parameters and running statistics (train mode)"""
    ref = unet_ref.UNet(3, 1, 32)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in m.localizer.state_dict().items()})
    return ref.train()


def _pred_on_same_input(ref_unet, attacked):
    """what the oracle UNet predicts from the SAME quantised input the HIP UNet saw.  (Comparing against a prediction made from the
    oracle's own `attacked` would mix in the flipped quantisation steps: in training mode one flipped input pixel moves the batch
    statistics of all 18 BatchNorm layers, i.e. every output pixel a little.)"""
    with torch.no_grad():
        return ref_unet(attacked.detach().cpu().float())


def _check_integer_mask(m, images):
    """integer tamper masks bit-exact (north_star): the eval-mode localiser on the HIP f32 path against the oracle UNet holding the
    SAME parameters and running statistics, thresholded at 0.5 -- identical wherever the oracle's sigmoid is not within 1e-4 of
    the threshold (f32 round-off of the two implementations)"""
    ref = unet_ref.UNet(3, 1, 32)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.localizer.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        rp = ref(images.cpu().float()).numpy()
    mk = m.localise_mask(images)
    assert mk.dtype == torch.uint8 and tuple(mk.shape) == tuple(rp.shape) and set(np.unique(mk.cpu().numpy())) <= {0, 1}
    safe = np.abs(rp - 0.5) > 1e-4
    assert safe.mean() > 0.99
    assert np.array_equal(mk.cpu().numpy()[safe], (rp > 0.5).astype(np.uint8)[safe])


def test_c5_clip_256_vs_oracle(tmp_path):
    """BASELINE configs[4], one rank's shard: a 16-frame 256x256 clip folded into the batch, JpegSS(70) attack, UNet head,
    gradient clipping -- f32 path at the north_star tolerance, the bf16 production path at its bound, and f16 + the device-side GradScaler
    (the dtype BASELINE configs[4] states: /root/reference/models/IRNcrop_model.py:340,407-416) at the f16 bound, against the CPU oracle."""
    S, T = 256, 16
    clip = detgen.uniform((1, 3, T, S, S), 9300)
    mask5 = torch.zeros(1, 1, T, S, S)
    mask5[..., 64:192, 32:160] = 1.0
    mask5[:, :, 3] = 0.0
    images = clip.permute(0, 2, 1, 3, 4).reshape(T, 3, S, S)
    mask = mask5.permute(0, 2, 1, 3, 4).reshape(T, 1, S, S)
    messages = detgen.bits((T, 30), 9301)
    previous = detgen.uniform((T, 3, S, S), 9302)
    attack = lambda x: jpeg_ref.jpeg_layer(x, 70, "ss")   # noqa: E731
    h = make_ref(S, attack)
    unet = detgen.fill_module(unet_ref.UNet(3, 1, 32))
    loc = localise_ref.LocaliseRef(h, unet, attack, gradient_clipping=1.0)
    rlogs, routs, _ = loc.step(images, messages, previous, mask)
    res = {}
    for name in ("f32", "bf16", "f16"):
        m = make_model(tmp_path, S, attacks=["JpegSS70"], gradient_clipping=1.0, compute_dtype=name)
        assert (m.amp is not None) == (name == "f16")
        m.previous_images = previous.cuda()
        m.previous_previous_images = previous.cuda()
        m.feed_data({"GT": clip, "mask": mask5, "messages": messages})
        assert m.real_H.shape == (T, 3, S, S) and m.mask.shape == (T, 1, S, S)
        unet_before = _unet_snapshot(m)
        logs, _ = m.optimize_parameters(1, None)
        d, o = dict(logs), m.last_outputs
        res[name] = dict(enc=rel(o["encoded"], routs["encoded"]), dec=rel(o["decoded"], routs["decoded"]),
                         pred=rel(o["pred"], _pred_on_same_input(unet_before, o["attacked"])),
                         lB=abs(d["lB"] - rlogs["lB"]), PF=abs(d["PF"] - rlogs["PF"]), loss=abs(d["loss"] - rlogs["loss"]) / max(1.0, abs(rlogs["loss"])),
                         tampered=float((np.abs(o["tampered"].cpu().numpy() - routs["tampered"].numpy()) > 1e-6).mean()))
        res[name]["pred_vs_oracle_run"] = float(np.abs(o["pred"].cpu().numpy() - routs["pred"].numpy()).max())
        if name == "f32":
            _check_integer_mask(m, images)    # integer tamper mask: bit-exact
    print("C5", res)
    f, b = res["f32"], res["bf16"]
    # (pred: against the oracle UNet on the SAME quantised input -- see _pred_on_same_input; against the oracle's own run, whose input
    # differs by a few flipped quantisation steps, a sanity bound)
    assert f["enc"] < 1e-3 and f["dec"] < 1e-3 and f["pred"] < 1e-3 and f["lB"] < 1e-3 and f["PF"] < 1e-2 and f["loss"] < 1e-3, f
    assert f["tampered"] < 1e-3 and f["pred_vs_oracle_run"] < 1e-1, f
    # bf16: the 18-conv UNet rounds its activations to bf16 per layer
    assert b["enc"] < 2e-2 and b["dec"] < 5e-2 and b["pred"] < 1.5e-1 and b["lB"] < 2e-2 and b["loss"] < 2e-2, b
    # f16 (C5's stated dtype): 2^-11 relative rounding per layer instead of 2^-9 -- the bounds of test_gpu_fp16.test_f16_step_vs_oracle on
    # the watermark path, a quarter of bf16's on the UNet's prediction (same-input oracle UNet) and the logged losses
    h16 = res["f16"]
    assert h16["enc"] < 5e-3 and h16["dec"] < 1.5e-2 and h16["pred"] < 4e-2 and h16["lB"] < 5e-3 and h16["loss"] < 5e-3, h16


def test_grad_sync_one_rank_is_identity():
    """the data-parallel code path (async bucket start / finish, 1/world folded into the optimiser kernel, the encoder's two
    buckets) on a one-rank RCCL group: bit-identical to grad_sync=None"""
    import os
    import socket
    import torch.distributed as dist
    from video_watermarking_forgery_detection_amd.distributed import GradSync
    from video_watermarking_forgery_detection_amd.noise_layers import JpegSS
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        images = detgen.uniform((4, 3, 64, 64), 9400)
        messages = detgen.bits((4, 30), 9401)
        out = []
        for sync in (None, GradSync(force=True)):
            h = make_hidden(64, JpegSS(50), torch.bfloat16)
            h.grad_sync = sync
            for _ in range(2):
                losses, (e, _, d) = h.train_on_batch([images, messages])
            out.append((dict(losses), e.clone(), d.clone(), [m.flat_params.clone() for _, m in nets(h)]))
        assert out[1][0] == out[0][0]
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
        for a, b in zip(out[0][3], out[1][3]):
            assert torch.equal(a, b)
        assert GradSync(force=True).active and not GradSync().active
    finally:
        dist.destroy_process_group()


def test_training_reduces_bit_error_bf16_like_f32():
    """a few hundred steps on random frames (tools/train_sanity.py as a test): the fused bf16 step learns like the exact-f32 step"""
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.noise_layers import Identity
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    dev = torch.device("cuda", 0)
    final = {}
    for dt in (torch.bfloat16, torch.float32):
        torch.manual_seed(10)
        h = Hidden(HiDDenConfiguration(H=64, W=64), dev, Identity(), None, compute_dtype=dt)
        first = None
        for it in range(400):
            images = torch.rand(16, 3, 64, 64, device=dev)
            messages = torch.randint(0, 2, (16, 30), device=dev).float()
            losses, _ = h.train_on_batch([images, messages])
            if it % 100 == 0 or it == 399:
                vals = dict(losses)
                assert all(v == v and abs(v) < 1e4 for v in vals.values()), vals
                first = first or vals
        final[dt] = vals
        assert vals["bitwise-error  "] < 0.40 < first["bitwise-error  "] + 0.15, (first, vals)
        assert vals["dec_mse        "] < first["dec_mse        "], (first, vals)
    print("train sanity", final)
    assert abs(final[torch.bfloat16]["bitwise-error  "] - final[torch.float32]["bitwise-error  "]) < 0.08
    assert abs(final[torch.bfloat16]["dec_mse        "] - final[torch.float32]["dec_mse        "]) < 0.03


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_dead_discriminator_grads_switch_changes_nothing_but_the_dead_grads(dt):
    """Hidden(keep_dead_discriminator_grads=False): the generator's pass through the discriminator computes the gradient wrt `encoded`
    only (/root/reference/hidden_models/hidden.py:85-103 accumulates weight gradients there that hidden.py:67 zeroes unread).  Two steps of
    both modes: identical losses, outputs and parameters of all three networks; the discriminator's left-over .grad differs."""
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.noise_layers import JpegSS
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    images = detgen.uniform((4, 3, 64, 64), 9700)
    messages = detgen.bits((4, 30), 9701)
    out = []
    for keep in (True, False):
        h = Hidden(HiDDenConfiguration(H=64, W=64), torch.device("cuda"), JpegSS(50), None, compute_dtype=dt, keep_dead_discriminator_grads=keep)
        for m in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
            detgen.fill_module(m)
        for _ in range(2):
            losses, (e, n, d) = h.train_on_batch([images, messages])
        out.append((dict(losses), e.clone(), n.clone(), d.clone(), [m.flat_params.clone() for _, m in nets(h)], h.discriminator.flat_grads.clone()))
    a, b = out
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for pa, pb in zip(a[4], b[4]):
        assert torch.equal(pa, pb)
    assert not torch.equal(a[5], b[5])       # the dead gradients: kept in the first mode, not computed in the second
