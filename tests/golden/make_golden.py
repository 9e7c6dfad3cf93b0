#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz from the REFERENCE.

Runs only in the build container (needs /root/reference, read-only, imported
unmodified with PYTHONDONTWRITEBYTECODE=1).  The reference never travels: what
is committed are the .npz files (inputs are re-derivable from seeds through
detgen.py, expected outputs are stored) plus this script.

Import shims: torchvision / cv2 / kornia are absent from the
image and are not on the hot path -> empty stub modules; `hidden_models/*`
imports `options.HiDDenConfiguration` and `model.conv_bn_relu`, which do not
exist in the reference tree -> a dataclass with the fields the modules read,
and an alias `model` -> `hidden_models`.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import dataclasses
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detgen  # noqa: E402

REF = "/root/reference"
warnings.filterwarnings("ignore")
torch.set_num_threads(8)


# --------------------------------------------------------------------------- shims
def install_shims():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    tvm = types.ModuleType("torchvision.models")     # imported, never called, by models/conditional_jpeg_generator.py:6
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.models"] = tvm
    sys.modules["cv2"] = types.ModuleType("cv2")
    k = types.ModuleType("kornia")
    kf = types.ModuleType("kornia.filters")

    class _Absent(nn.Module):
        def __init__(self, *a, **kw):
            super().__init__()

        def forward(self, x):
            raise RuntimeError("kornia is absent from this image")

    kf.MedianBlur = _Absent
    kf.GaussianBlur2d = _Absent
    k.filters = kf
    sys.modules["kornia"] = k
    sys.modules["kornia.filters"] = kf

    # options.HiDDenConfiguration: fields read at encoder.py:13-25, decoder.py:15-22,
    # discriminator.py:12-18, hidden.py:27,99-100
    import options as ref_options

    @dataclasses.dataclass
    class HiDDenConfiguration:
        H: int
        W: int
        message_length: int = 30
        encoder_blocks: int = 4
        encoder_channels: int = 64
        decoder_blocks: int = 7
        decoder_channels: int = 64
        use_discriminator: bool = True
        use_vgg: bool = False
        discriminator_blocks: int = 3
        discriminator_channels: int = 64
        decoder_loss: float = 1.0
        encoder_loss: float = 0.7
        adversarial_loss: float = 1e-3

    ref_options.HiDDenConfiguration = HiDDenConfiguration
    import hidden_models
    import hidden_models.conv_bn_relu

    sys.modules["model"] = hidden_models
    sys.modules["model.conv_bn_relu"] = hidden_models.conv_bn_relu
    return HiDDenConfiguration


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


# --------------------------------------------------------------------------- block JPEG (noise_layers/jpeg.py)
def gen_jpeg(out):
    from noise_layers.jpeg import Jpeg, JpegSS, JpegMask

    kinds = {"Jpeg": Jpeg, "JpegSS": JpegSS, "JpegMask": JpegMask}
    # (H, W) all pad to a square multiple of 8 -- the only shapes for which the
    # reference's split/cat block reshuffle (jpeg.py:123-127) is self-consistent.
    sizes = [(16, 16), (64, 64), (30, 27)]
    seed = 100
    for kname, K in kinds.items():
        for Q in (10, 50, 75, 90):
            for (H, W) in sizes:
                for sub in (0, 2):
                    if (H, W) == (64, 64) and Q in (10, 75) and sub == 2:
                        continue  # keep the fixture small
                    seed += 1
                    x = detgen.uniform((2, 3, H, W), seed).requires_grad_(True)
                    g = detgen.normal((2, 3, H, W), seed + 5000)
                    layer = K(Q, subsample=sub)
                    y = layer(x)
                    (y * g).sum().backward()
                    key = f"{kname}_Q{Q}_{H}x{W}_s{sub}"
                    out[key + "/seed"] = np.int64(seed)
                    out[key + "/y"] = npy(y)
                    out[key + "/gx"] = npy(x.grad)
                    out[key + "/name"] = np.array(layer.name)
    # one block's DCT coefficients (a7: yuv_dct) for a known-answer test
    x = detgen.uniform((1, 3, 8, 8), 4242)
    layer = Jpeg(50)
    coef, pw, ph = layer.yuv_dct(x, 0)
    out["dct_block/x_seed"] = np.int64(4242)
    out["dct_block/coef"] = npy(coef)
    q = layer.std_quantization(coef, layer.scale_factor)
    out["dct_block/q50"] = npy(q)


# --------------------------------------------------------------------------- DiffJPEG stages (utils/JPEG.py)
def gen_diffjpeg(out):
    import utils.JPEG as RJ

    seed = 300
    for quality in (50, 75, 90):
        for (H, W) in ((32, 32), (64, 64), (32, 48)):
            for rname, rfn in (("round", torch.round), ("round0", RJ.round_only_at_0)):
                seed += 1
                factor = RJ.quality_to_factor(quality)
                comp = RJ.compress_jpeg(rounding=rfn, factor=factor)
                dec = RJ.decompress_jpeg(H, W, rounding=rfn, factor=factor)
                x = detgen.uniform((2, 3, H, W), seed).requires_grad_(True)
                g = detgen.normal((2, 3, H, W), seed + 5000)
                y, cb, cr = comp(x)
                rec = dec(y, cb, cr)
                (rec * g).sum().backward()
                key = f"DiffJPEG_q{quality}_{H}x{W}_{rname}"
                out[key + "/seed"] = np.int64(seed)
                out[key + "/y"] = npy(y)
                out[key + "/cb"] = npy(cb)
                out[key + "/cr"] = npy(cr)
                out[key + "/rec"] = npy(rec)
                out[key + "/gx"] = npy(x.grad)
    out["quality_to_factor"] = np.array([[q, RJ.quality_to_factor(q)] for q in (10, 25, 50, 75, 90, 95)], dtype=np.float64)
    xs = torch.linspace(-2.0, 2.0, 81)
    out["round_only_at_0/x"] = npy(xs)
    out["round_only_at_0/y"] = npy(RJ.round_only_at_0(xs))
    out["diff_round/y"] = npy(RJ.diff_round(xs))


# --------------------------------------------------------------------------- stencil / resample attacks
def gen_attacks(out):
    from noise_layers.gaussian_blur import GaussianBlur
    from noise_layers.resize import Resize
    from noise_layers.crop import Crop
    from noise_layers.combined import Combined
    from noise_layers.identity import Identity
    from noise_layers.jpeg import Jpeg, JpegSS, JpegMask
    from models.modules.Quantization import Quantization

    # a11: GaussianBlur.forward hard-codes .cuda() (gaussian_blur.py:55); use its
    # own kernel builder and apply the returned depthwise conv on CPU.
    gb = GaussianBlur()
    conv = gb.get_gaussian_kernel(channels=3)
    out["gauss/kernel"] = npy(conv.weight[0, 0])
    for i, (H, W) in enumerate(((16, 16), (33, 47), (64, 64))):
        x = detgen.uniform((2, 3, H, W), 500 + i).requires_grad_(True)
        g = detgen.normal((2, 3, H, W), 5500 + i)
        y = conv(x)
        (y * g).sum().backward()
        out[f"gauss_{H}x{W}/seed"] = np.int64(500 + i)
        out[f"gauss_{H}x{W}/y"] = npy(y)
        out[f"gauss_{H}x{W}/gx"] = npy(x.grad)

    # a13: Resize with explicit ratio
    rs = Resize()
    i = 0
    for (H, W) in ((32, 32), (48, 40)):
        for r in (0.5, 0.7, 1.3):
            i += 1
            x = detgen.uniform((2, 3, H, W), 600 + i).requires_grad_(True)
            g = detgen.normal((2, 3, H, W), 5600 + i)
            y = rs(x, resize_ratio=r)
            (y * g).sum().backward()
            key = f"resize_{H}x{W}_r{r}"
            out[key + "/seed"] = np.int64(600 + i)
            out[key + "/y"] = npy(y)
            out[key + "/gx"] = npy(x.grad)

    # a14: Crop with explicit apex
    cr = Crop()
    i = 0
    for (H, W), apex in (((32, 32), (4, 28, 2, 26)), ((64, 64), (8, 56, 8, 56)), ((40, 48), (0, 31, 5, 48))):
        i += 1
        x = detgen.uniform((2, 3, H, W), 700 + i).requires_grad_(True)
        g = detgen.normal((2, 3, H, W), 5700 + i)
        np.random.seed(0)
        y, ap = cr(x, apex=apex)
        (y * g).sum().backward()
        key = f"crop_{H}x{W}_{'_'.join(map(str, apex))}"
        out[key + "/seed"] = np.int64(700 + i)
        out[key + "/apex"] = np.array(ap, dtype=np.int64)
        out[key + "/y"] = npy(y)
        out[key + "/gx"] = npy(x.grad)
    # random-apex bookkeeping (crop.py:13-30,33-40): numpy-seeded draws
    for s in (1, 2, 3):
        np.random.seed(s)
        x = detgen.uniform((1, 3, 32, 32), 750 + s)
        y, ap = cr(x)
        out[f"crop_rand_seed{s}/apex"] = np.array(ap, dtype=np.int64)
        out[f"crop_rand_seed{s}/y"] = npy(y)

    # a16: Quantization
    qz = Quantization()
    x = detgen.uniform((2, 3, 16, 16), 800, lo=-0.2, hi=1.2).requires_grad_(True)
    g = detgen.normal((2, 3, 16, 16), 5800)
    y = qz(x)
    (y * g).sum().backward()
    out["quant/seed"] = np.int64(800)
    out["quant/y"] = npy(y)
    out["quant/gx"] = npy(x.grad)

    # a9 / a15: Combined dispatch + names
    comb = Combined([JpegMask(80), Jpeg(80), JpegSS(70), Identity()])
    x = detgen.uniform((1, 3, 16, 16), 900)
    names = []
    for k in range(4):
        y = comb(x, id=k)
        names.append(comb.name)
        out[f"combined_id{k}/y"] = npy(y)
    out["combined/names"] = np.array(names)
    out["combined/seed"] = np.int64(900)


# --------------------------------------------------------------------------- HiDDeN nets
def gen_hidden(out, Cfg):
    from hidden_models.conv_bn_relu import ConvBNRelu
    from hidden_models.encoder import Encoder
    from hidden_models.decoder import Decoder
    from hidden_models.discriminator import Discriminator

    # a1
    for (cin, cout) in ((3, 64), (64, 64), (64, 30)):
        m = detgen.fill_module(ConvBNRelu(cin, cout)).train()
        x = detgen.normal((2, cin, 16, 16), 1000 + cin + cout).requires_grad_(True)
        g = detgen.normal((2, cout, 16, 16), 6000 + cin + cout)
        y = m(x)
        (y * g).sum().backward()
        key = f"cbr_{cin}_{cout}"
        out[key + "/seed"] = np.int64(1000 + cin + cout)
        out[key + "/y"] = npy(y)
        out[key + "/gx"] = npy(x.grad)
        out[key + "/gw"] = npy(m.layers[0].weight.grad)
        out[key + "/gb"] = npy(m.layers[0].bias.grad)
        out[key + "/ggamma"] = npy(m.layers[1].weight.grad)
        out[key + "/gbeta"] = npy(m.layers[1].bias.grad)
        out[key + "/running_mean"] = npy(m.layers[1].running_mean)
        out[key + "/running_var"] = npy(m.layers[1].running_var)

    # a2-a4 at 2x3x32x32
    cfg = Cfg(H=32, W=32)
    enc = detgen.fill_module(Encoder(cfg)).train()
    dec = detgen.fill_module(Decoder(cfg)).train()
    dis = detgen.fill_module(Discriminator(cfg)).train()
    img = detgen.uniform((2, 3, 32, 32), 1100).requires_grad_(True)
    msg = detgen.bits((2, 30), 1101)
    e = enc(img, msg.clone())
    ge = detgen.normal((2, 3, 32, 32), 6100)
    (e * ge).sum().backward()
    out["enc32/y"] = npy(e)
    out["enc32/gimg"] = npy(img.grad)
    for n, p in enc.named_parameters():
        out[f"enc32/g/{n}"] = npy(detgen.subsample(p.grad, 97))
    x = detgen.uniform((2, 3, 32, 32), 1102).requires_grad_(True)
    d = dec(x)
    gd = detgen.normal((2, 30), 6101)
    (d * gd).sum().backward()
    out["dec32/y"] = npy(d)
    out["dec32/gx"] = npy(x.grad)
    for n, p in dec.named_parameters():
        out[f"dec32/g/{n}"] = npy(detgen.subsample(p.grad, 97))
    x = detgen.uniform((2, 3, 32, 32), 1103).requires_grad_(True)
    d = dis(x)
    gd = detgen.normal((2, 1), 6102)
    (d * gd).sum().backward()
    out["dis32/y"] = npy(d)
    out["dis32/gx"] = npy(x.grad)
    for n, p in dis.named_parameters():
        out[f"dis32/g/{n}"] = npy(detgen.subsample(p.grad, 97))
    out["param_counts"] = np.array([sum(p.numel() for p in m.parameters()) for m in (enc, dec, dis)], dtype=np.int64)

    # config C1: encoder -> identity -> decoder, single 128x128 frame
    cfg = Cfg(H=128, W=128)
    enc = detgen.fill_module(Encoder(cfg)).train()
    dec = detgen.fill_module(Decoder(cfg)).train()
    img = detgen.uniform((1, 3, 128, 128), 1200)
    msg = detgen.bits((1, 30), 1201)
    e = enc(img, msg.clone())
    d = dec(e)
    out["c1/encoded_sub"] = npy(detgen.subsample(e, 13))
    out["c1/encoded_mean_abs"] = np.float64(e.abs().mean().item())
    out["c1/decoded"] = npy(d)


# --------------------------------------------------------------------------- full GAN step (hidden.py:54-118 order)
def gen_step(out, Cfg):
    from hidden_models.encoder import Encoder
    from hidden_models.decoder import Decoder
    from hidden_models.discriminator import Discriminator
    from noise_layers.jpeg import Jpeg, JpegSS, JpegMask
    from noise_layers.identity import Identity

    for nname, noise in (("JpegSS50", JpegSS(50)), ("Jpeg50", Jpeg(50)), ("JpegMask50", JpegMask(50)), ("Identity", Identity())):
        cfg = Cfg(H=32, W=32)
        enc = detgen.fill_module(Encoder(cfg)).train()
        dec = detgen.fill_module(Decoder(cfg)).train()
        dis = detgen.fill_module(Discriminator(cfg)).train()
        # hidden.py:24-25 -- one Adam over encoder_decoder.parameters() (encoder then decoder), one over D
        opt_ed = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()))
        opt_d = torch.optim.Adam(dis.parameters())
        bce = nn.BCEWithLogitsLoss()
        mse = nn.MSELoss()
        images = detgen.uniform((4, 3, 32, 32), 2000)
        messages = detgen.bits((4, 30), 2001)
        B = 4
        for it in range(2):
            opt_d.zero_grad()
            t1 = torch.full((B, 1), 1.0)
            t0 = torch.full((B, 1), 0.0)
            d_cover = dis(images)
            l_dc = bce(d_cover, t1)
            l_dc.backward()
            encoded = enc(images, messages.clone())
            noised = noise(encoded)
            decoded = dec(noised)
            d_enc = dis(encoded.detach())
            l_de = bce(d_enc, t0)
            l_de.backward()
            if it == 0:
                for n, p in dis.named_parameters():
                    out[f"step_{nname}/gD/{n}"] = npy(detgen.subsample(p.grad, 31))
            opt_d.step()
            opt_ed.zero_grad()
            d_enc2 = dis(encoded)
            l_adv = bce(d_enc2, t1)
            l_enc = mse(encoded, images)
            l_dec = mse(decoded, messages)
            g_loss = cfg.adversarial_loss * l_adv + cfg.encoder_loss * l_enc + cfg.decoder_loss * l_dec
            g_loss.backward()
            if it == 0:
                for n, p in enc.named_parameters():
                    out[f"step_{nname}/gE/{n}"] = npy(detgen.subsample(p.grad, 31))
                for n, p in dec.named_parameters():
                    out[f"step_{nname}/gDec/{n}"] = npy(detgen.subsample(p.grad, 31))
                out[f"step_{nname}/encoded"] = npy(encoded)
                out[f"step_{nname}/noised"] = npy(noised)
                out[f"step_{nname}/decoded"] = npy(decoded)
            opt_ed.step()
            dr = decoded.detach().numpy().round().clip(0, 1)
            biterr = np.sum(np.abs(dr - messages.numpy())) / (B * 30)
            out[f"step_{nname}/losses_it{it}"] = np.array(
                [g_loss.item(), l_enc.item(), l_dec.item(), biterr, l_adv.item(), l_dc.item(), l_de.item()], dtype=np.float64)
        for n, p in list(enc.state_dict().items()):
            out[f"step_{nname}/wE/{n}"] = npy(detgen.subsample(p.float(), 31))
        for n, p in list(dec.state_dict().items()):
            out[f"step_{nname}/wDec/{n}"] = npy(detgen.subsample(p.float(), 31))
        for n, p in list(dis.state_dict().items()):
            out[f"step_{nname}/wD/{n}"] = npy(detgen.subsample(p.float(), 31))


# --------------------------------------------------------------------------- UNet
def gen_unet(out):
    from network.UNet import UNet

    net = detgen.fill_module(UNet(3, 1, 32)).train()
    out["param_count"] = np.int64(sum(p.numel() for p in net.parameters()))
    for (B, H) in ((1, 32), (2, 64)):
        net.zero_grad()
        x = detgen.uniform((B, 3, H, H), 3000 + H).requires_grad_(True)
        g = detgen.normal((B, 1, H, H), 8000 + H)
        y = net(x)
        (y * g).sum().backward()
        key = f"unet_{B}x{H}"
        out[key + "/seed"] = np.int64(3000 + H)
        out[key + "/y"] = npy(y)
        out[key + "/gx"] = npy(x.grad)
        for n, p in net.named_parameters():
            out[f"{key}/g/{n}"] = npy(detgen.subsample(p.grad, 997))
            out[f"{key}/gnorm/{n}"] = np.float64(p.grad.norm().item())
        # reset running stats so the second case starts from a fresh module too
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.reset_running_stats()


# --------------------------------------------------------------------------- config C3: the HiDDeN-order step under the stencil / resample attacks
def _ref_attacks():
    """name -> callable image -> image built from the REFERENCE's noise layers (the C3 attack cycle of SURVEY §8d).  GaussianBlur's
    forward hard-codes .cuda() (gaussian_blur.py:55): its own kernel builder is applied on CPU.  MiddleBlur is kornia (absent): the
    fixture uses the build's definition (oracle/attacks_ref.median_blur, parity unpinned for that one op)."""
    from noise_layers.gaussian_blur import GaussianBlur
    from noise_layers.resize import Resize
    from noise_layers.crop import Crop
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import attacks_ref
    gconv = GaussianBlur().get_gaussian_kernel(channels=3)
    rs, cr = Resize(), Crop()

    def crop_fn(x):
        H, W = x.shape[2], x.shape[3]
        return cr(x, apex=(H // 8, H // 8 + int(0.75 * H), W // 8, W // 8 + int(0.75 * W)))[0]

    return {"GaussianBlur": lambda x: gconv(x), "Resize0.7": lambda x: rs(x, resize_ratio=0.7), "Crop0.75": crop_fn,
            "MiddleBlur3": lambda x: attacks_ref.median_blur(x, 3)}


def _hidden_step(out, prefix, noise, Cfg, size=32, B=4, iters=2, stride=31):
    """one HiDDeN-order GAN step (hidden.py:54-118) on the reference's Encoder / Decoder / Discriminator, like gen_step"""
    from hidden_models.encoder import Encoder
    from hidden_models.decoder import Decoder
    from hidden_models.discriminator import Discriminator
    cfg = Cfg(H=size, W=size)
    enc = detgen.fill_module(Encoder(cfg)).train()
    dec = detgen.fill_module(Decoder(cfg)).train()
    dis = detgen.fill_module(Discriminator(cfg)).train()
    opt_ed = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()))
    opt_d = torch.optim.Adam(dis.parameters())
    bce, mse = nn.BCEWithLogitsLoss(), nn.MSELoss()
    images = detgen.uniform((B, 3, size, size), 2000)
    messages = detgen.bits((B, 30), 2001)
    t1, t0 = torch.full((B, 1), 1.0), torch.full((B, 1), 0.0)
    for it in range(iters):
        opt_d.zero_grad()
        l_dc = bce(dis(images), t1)
        l_dc.backward()
        encoded = enc(images, messages.clone())
        noised = noise(encoded)
        decoded = dec(noised)
        l_de = bce(dis(encoded.detach()), t0)
        l_de.backward()
        if it == 0:
            for n, p in dis.named_parameters():
                out[f"{prefix}/gD/{n}"] = npy(detgen.subsample(p.grad, stride))
        opt_d.step()
        opt_ed.zero_grad()
        l_adv = bce(dis(encoded), t1)
        l_enc = mse(encoded, images)
        l_dec = mse(decoded, messages)
        g_loss = cfg.adversarial_loss * l_adv + cfg.encoder_loss * l_enc + cfg.decoder_loss * l_dec
        g_loss.backward()
        if it == 0:
            for n, p in enc.named_parameters():
                out[f"{prefix}/gE/{n}"] = npy(detgen.subsample(p.grad, stride))
            for n, p in dec.named_parameters():
                out[f"{prefix}/gDec/{n}"] = npy(detgen.subsample(p.grad, stride))
            out[f"{prefix}/encoded"] = npy(encoded)
            out[f"{prefix}/noised"] = npy(noised)
            out[f"{prefix}/decoded"] = npy(decoded)
        opt_ed.step()
        dr = decoded.detach().numpy().round().clip(0, 1)
        biterr = np.sum(np.abs(dr - messages.numpy())) / (B * 30)
        out[f"{prefix}/losses_it{it}"] = np.array(
            [g_loss.item(), l_enc.item(), l_dec.item(), biterr, l_adv.item(), l_dc.item(), l_de.item()], dtype=np.float64)
    for tag, m in (("wE", enc), ("wDec", dec), ("wD", dis)):
        for n, p in list(m.state_dict().items()):
            out[f"{prefix}/{tag}/{n}"] = npy(detgen.subsample(p.float(), stride))


def gen_step_c3(out, Cfg):
    for name, fn in _ref_attacks().items():
        _hidden_step(out, f"step_{name}", fn, Cfg)


# --------------------------------------------------------------------------- tamper-localisation branch (a18), composed from the reference's modules
def gen_localise(out, Cfg):
    """The composition models/IRNcrop_model.py:337-416 describes (it cannot run as written, SURVEY box) around the HiDDeN-order
    step, built from the REFERENCE's modules: hidden_models.{Encoder,Decoder,Discriminator}, network.UNet, noise layers,
    models.modules.Quantization, metrics.PSNR; clamp_with_grad / postprocess are the three-line helpers of IRNcrop_model.py:320-322,
    660-664 restated.  Same order as oracle/localise_ref.py."""
    from hidden_models.encoder import Encoder
    from hidden_models.decoder import Decoder
    from hidden_models.discriminator import Discriminator
    from network.UNet import UNet
    from models.modules.Quantization import Quantization
    from noise_layers.jpeg import JpegSS
    from metrics import PSNR

    def clamp_with_grad(t):
        return t + (torch.clamp(t, 0, 1) - t).clone().detach()

    def postprocess(img):
        return (img * 255.0).permute(0, 2, 3, 1).int()

    attacks = dict(_ref_attacks())
    jss = JpegSS(70)
    attacks["JpegSS70"] = lambda x: jss(x)
    quant = Quantization()
    psnr = PSNR(255.0)
    size, B = 32, 4
    for case, (aname, clip, enc_gain) in {"jpegss_clip": ("JpegSS70", 1.0, 1.0), "resize": ("Resize0.7", None, 1.0),
                                          "gauss_hipsnr": ("GaussianBlur", 0.5, 0.02)}.items():
        attack = attacks[aname]
        cfg = Cfg(H=size, W=size)
        enc = detgen.fill_module(Encoder(cfg)).train()
        dec = detgen.fill_module(Decoder(cfg)).train()
        dis = detgen.fill_module(Discriminator(cfg)).train()
        unet = detgen.fill_module(UNet(3, 1, 32)).train()
        images = detgen.uniform((B, 3, size, size), 2100)
        if enc_gain != 1.0:
            # a near-identity embedder (final 1x1 conv ~ 0, its bias = a constant): PSNR above 33 dB needs encoded ~ images, which a
            # random-init encoder never gives; instead the "cover" is made the constant image the encoder emits, so the gate's
            # 0.8 branch (IRNcrop_model.py:387-388) is exercised
            with torch.no_grad():
                enc.final_layer.weight.mul_(enc_gain)
                enc.final_layer.bias.copy_(torch.tensor([0.4, 0.5, 0.6]))
            images = torch.tensor([0.4, 0.5, 0.6]).view(1, 3, 1, 1).expand(B, 3, size, size).contiguous() + 0.01 * (images - 0.5)
        messages = detgen.bits((B, 30), 2101)
        previous = detgen.uniform((B, 3, size, size), 2102)
        mask = torch.zeros(B, 1, size, size)
        mask[:, :, 8:24, 4:20] = 1.0
        mask[1, :, :, :] = 0.0
        opt_ed = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()))
        opt_d = torch.optim.Adam(dis.parameters())
        opt_u = torch.optim.Adam(unet.parameters())
        bce, mse = nn.BCEWithLogitsLoss(), nn.MSELoss()
        t1, t0 = torch.full((B, 1), 1.0), torch.full((B, 1), 0.0)
        key = f"loc_{case}"
        out[key + "/attack"] = np.array(aname)
        out[key + "/clip"] = np.float64(clip or 0.0)
        out[key + "/enc_gain"] = np.float64(enc_gain)
        for it in range(2):
            opt_d.zero_grad()
            l_dc = bce(dis(images), t1)
            l_dc.backward()
            encoded = enc(images, messages.clone())
            noised = attack(encoded)
            decoded = dec(noised)
            l_de = bce(dis(encoded.detach()), t0)
            l_de.backward()
            if clip:
                nn.utils.clip_grad_norm_(dis.parameters(), clip)
            opt_d.step()
            opt_ed.zero_grad()
            opt_u.zero_grad()
            l_adv = bce(dis(encoded), t1)
            l_enc = mse(encoded, images)
            l_dec = mse(decoded, messages)
            fwd_img = quant(clamp_with_grad(encoded))
            pf = float(psnr(postprocess(images), postprocess(fwd_img)))
            gate = 1.0 if pf < 33 else 0.8
            tampered = fwd_img * (1 - mask) + previous * mask
            attacked = quant(clamp_with_grad(attack(tampered)))
            pred = unet(attacked)
            l_loc = bce(pred, mask)
            loss = cfg.adversarial_loss * l_adv + gate * cfg.encoder_loss * l_enc + cfg.decoder_loss * l_dec + 1.0 * l_loc
            loss.backward()
            if clip:
                nn.utils.clip_grad_norm_(list(enc.parameters()) + list(dec.parameters()), clip)
                nn.utils.clip_grad_norm_(unet.parameters(), clip)
            if it == 0:
                for tag, m, st in (("gE", enc, 31), ("gDec", dec, 31), ("gU", unet, 997)):
                    for n, p in m.named_parameters():
                        out[f"{key}/{tag}/{n}"] = npy(detgen.subsample(p.grad, st))
                        out[f"{key}/{tag}norm/{n}"] = np.float64(p.grad.norm().item())
                for nm, t in (("encoded", encoded), ("tampered", tampered), ("attacked", attacked), ("pred", pred), ("decoded", decoded)):
                    out[f"{key}/{nm}"] = npy(t)
            opt_ed.step()
            opt_u.step()
            g3 = cfg.adversarial_loss * l_adv + cfg.encoder_loss * l_enc + cfg.decoder_loss * l_dec
            out[f"{key}/logs_it{it}"] = np.array([g3.item(), l_enc.item(), l_dec.item(), l_adv.item(), l_dc.item(), l_de.item(),
                                                  pf, gate, l_loc.item()], dtype=np.float64)
        for tag, m, st in (("wE", enc, 31), ("wDec", dec, 31), ("wD", dis, 31), ("wU", unet, 997)):
            for n, p in list(m.state_dict().items()):
                out[f"{key}/{tag}/{n}"] = npy(detgen.subsample(p.float(), st))


# --------------------------------------------------------------------------- TensorBoard scalar log written by the reference's own run
# --------------------------------------------------------------------------- row f1: Discriminator / FBCNN / QF_predictor
def _store_grads(out, key, net, stride=997):
    for n, p in net.named_parameters():
        if p.grad is None:
            out[f"{key}/nograd/{n}"] = np.int64(1)
            continue
        out[f"{key}/g/{n}"] = npy(detgen.subsample(p.grad, stride))
        out[f"{key}/gnorm/{n}"] = np.float64(p.grad.norm().item())


def gen_f1(out):
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self      # QF_predictor.__init__ moves its 5x5 masks with .cuda() (:732-736); CPU here
    try:
        from models.networks import Discriminator
        from models.conditional_jpeg_generator import FBCNN, QF_predictor, symm_pad

        # ---- Discriminator (networks.py:631-749)
        net = detgen.fill_f1(Discriminator(in_channels=3, use_sigmoid=True)).train()
        out["disc/keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
        x = detgen.uniform((2, 3, 64, 64), 9100).requires_grad_(True)
        y = net(x)
        g = detgen.normal(tuple(y.shape), 9101)
        (y * g).sum().backward()
        out["disc/y"], out["disc/gx"] = npy(y), npy(x.grad)
        _store_grads(out, "disc", net)
        for k, v in net.state_dict().items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                out[f"disc/after/{k}"] = npy(detgen.subsample(v, 7))
        net.eval()
        with torch.no_grad():
            out["disc/y_eval"] = npy(net(x))

        # ---- FBCNN (conditional_jpeg_generator.py:202-374); 36x44 exercises the replication padding to 40x48
        net = detgen.fill_f1(FBCNN(nc=[16, 32, 48, 64], nb=2)).train()
        out["fbcnn/keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
        x = detgen.uniform((2, 3, 36, 44), 9200).requires_grad_(True)
        qf = detgen.uniform((2, 1), 9201).requires_grad_(True)
        y, feats = net(x, qf)
        loss = (y * detgen.normal(tuple(y.shape), 9202)).sum()
        for i, f in enumerate(feats):
            loss = loss + 0.1 * (f * detgen.normal(tuple(f.shape), 9210 + i)).sum()
            out[f"fbcnn/feat{i}"] = npy(detgen.subsample(f, 13))
            out[f"fbcnn/feat{i}_shape"] = np.array(f.shape)
        loss.backward()
        out["fbcnn/y"], out["fbcnn/gx"], out["fbcnn/gqf"] = npy(y), npy(x.grad), npy(qf.grad)
        _store_grads(out, "fbcnn", net, 97)

        # ---- QF_predictor (:697-826)
        net = detgen.fill_f1(QF_predictor(nc=[16, 32, 48, 64], nb=2, classes=5)).train()
        out["qfp/keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
        x = detgen.uniform((2, 3, 32, 32), 9300).requires_grad_(True)
        bayar, qf = net(x)
        ((qf * detgen.normal(tuple(qf.shape), 9301)).sum() + 0.05 * (bayar * detgen.normal(tuple(bayar.shape), 9302)).sum()).backward()
        out["qfp/bayar"], out["qfp/qf"], out["qfp/gx"] = npy(bayar), npy(qf), npy(x.grad)
        out["qfp/bayar_weight_after"] = npy(net.BayarConv2D.weight)
        _store_grads(out, "qfp", net, 97)

        # ---- symm_pad (:865-885), pads larger than a trivial case
        im = detgen.uniform((1, 2, 5, 7), 9400)
        out["sympad/y"] = npy(symm_pad(im, (2, 3, 4, 1)))
    finally:
        torch.Tensor.cuda = _cuda


# --------------------------------------------------------------------------- row f2: the invertible embedder
def gen_f2(out):
    from models.invertible_net import Inveritible_Decolorization_PAMI, ResBlock, DenseBlock

    # the configuration models/IRNcrop_model.py:132-134 builds (4 channels, block_num [1,1,1], ResBlock subnets), at 32x32
    net = detgen.fill_f2(Inveritible_Decolorization_PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock)).train()
    out["pami/keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
    x = detgen.uniform((2, 4, 32, 32), 9500).requires_grad_(True)
    y = net(x)
    (y * detgen.normal(tuple(y.shape), 9501)).sum().backward()
    out["pami/y"], out["pami/gx"] = npy(y), npy(x.grad)
    _store_grads(out, "pami", net, 97)
    net.zero_grad()
    z = detgen.uniform((2, 4, 32, 32), 9502).requires_grad_(True)
    r, mid = net(z, rev=True)
    ((r * detgen.normal(tuple(r.shape), 9503)).sum() + 0.1 * (mid * detgen.normal(tuple(mid.shape), 9504)).sum()).backward()
    out["pami_rev/y"], out["pami_rev/mid"], out["pami_rev/gx"] = npy(r), npy(mid), npy(z.grad)
    _store_grads(out, "pami_rev", net, 97)
    with torch.no_grad():
        back, _ = net(net(x), rev=True)
    out["pami/roundtrip_err"] = np.float64((back - x).abs().max().item())

    # DenseBlock subnets, two levels, 3 channels
    net = detgen.fill_f2(Inveritible_Decolorization_PAMI(dims_in=[[3, 16, 16]], down_num=2, block_num=[1, 1], subnet_constructor=DenseBlock)).train()
    out["dense/keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
    x = detgen.uniform((2, 3, 16, 16), 9600).requires_grad_(True)
    y = net(x)
    (y * detgen.normal(tuple(y.shape), 9601)).sum().backward()
    out["dense/y"], out["dense/gx"] = npy(y), npy(x.grad)
    _store_grads(out, "dense", net, 97)


# --------------------------------------------------------------------------- rows f1 / f2: the two constructor options round 3 added
def gen_f12x(out):
    """HaarDownsampling(order_by_wavelet=True) (invertible_net.py:207-233) and QF_predictor(crop_pred=True) (conditional_jpeg_generator.py
    :772-784, :817-821); a file of its own so that f1.npz / f2.npz stay byte for byte what they were"""
    from models.invertible_net import HaarDownsampling
    net = HaarDownsampling([[3, 8, 12]], order_by_wavelet=True, rebalance=0.7)
    x = detgen.uniform((2, 3, 8, 12), 9700).requires_grad_(True)
    y = net(x)
    (y * detgen.normal(tuple(y.shape), 9701)).sum().backward()
    out["haarw/y"], out["haarw/gx"] = npy(y), npy(x.grad)
    z = detgen.uniform((2, 12, 4, 6), 9702).requires_grad_(True)
    r = net(z, rev=True)
    (r * detgen.normal(tuple(r.shape), 9703)).sum().backward()
    out["haarw/rev"], out["haarw/rev_gx"] = npy(r), npy(z.grad)

    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        from models.conditional_jpeg_generator import QF_predictor
        net = detgen.fill_f1(QF_predictor(nc=[16, 32, 48, 64], nb=2, classes=5, crop_pred=True)).train()
        out["qfpc/keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items()])
        x = detgen.uniform((2, 3, 32, 32), 9800).requires_grad_(True)
        img, qf = net(x)
        wimg = detgen.normal((2, 1, 64, 64), 9802).repeat_interleave(8, 2).repeat_interleave(8, 3)     # a 512 x 512 weight field from 64 x 64 draws
        ((qf * detgen.normal(tuple(qf.shape), 9801)).sum() + 0.05 * (img * wimg).sum()).backward()
        out["qfpc/img_shape"] = np.array(img.shape)
        out["qfpc/img_sub"] = npy(img[:, :, ::7, ::5])
        out["qfpc/img_sum"], out["qfpc/img_abs"] = np.float64(img.double().sum().item()), np.float64(img.double().abs().sum().item())
        out["qfpc/qf"], out["qfpc/gx"] = npy(qf), npy(x.grad)
        _store_grads(out, "qfpc", net, 97)
    finally:
        torch.Tensor.cuda = _cuda


def gen_tfevents():
    """the first records of one of the reference's TensorBoard event files (runs/RHI3: `PSNR Forward` ... scalars written through
    torch.utils.tensorboard at models/IRNcrop_model.py:399-400) -- DATA, kept as a known-answer file for utils/tb_writer.read_events and
    for the framing / crc / protobuf layout utils/tb_writer.SummaryWriter has to produce"""
    import struct
    src = os.path.join(REF, "runs", "RHI3", "events.out.tfevents.1653714410.group2.1832315.1")
    data = open(src, "rb").read()
    i, n = 0, 0
    while n < 10 and i + 12 <= len(data):
        (ln,) = struct.unpack("<Q", data[i:i + 8])
        i += 16 + ln
        n += 1
    out = os.path.join(HERE, "tfevents_head.bin")
    open(out, "wb").write(data[:i])
    print(f"tfevents: {n} records, {i} bytes -> {out}")


def main():
    if sys.argv[1:] == ["tfevents"]:
        return gen_tfevents()
    Cfg = install_shims()
    jobs = {
        "jpeg": lambda o: gen_jpeg(o),
        "diffjpeg": lambda o: gen_diffjpeg(o),
        "attacks": lambda o: gen_attacks(o),
        "hidden": lambda o: gen_hidden(o, Cfg),
        "step": lambda o: gen_step(o, Cfg),
        "unet": lambda o: gen_unet(o),
        "step_c3": lambda o: gen_step_c3(o, Cfg),
        "localise": lambda o: gen_localise(o, Cfg),
        "f1": lambda o: gen_f1(o),
        "f2": lambda o: gen_f2(o),
        "f12x": lambda o: gen_f12x(o),
    }
    which = sys.argv[1:] or list(jobs)
    for name in which:
        out = {}
        jobs[name](out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays -> {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
