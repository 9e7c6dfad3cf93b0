"""Oracle (test infrastructure): HiDDeN encoder / decoder / discriminator and the
GAN training step, restated in plain PyTorch-CPU fp32 with the reference's
state_dict keys.

  ConvBNRelu        /root/reference/hidden_models/conv_bn_relu.py:3-18
  Encoder           /root/reference/hidden_models/encoder.py:7-43
  Decoder           /root/reference/hidden_models/decoder.py:6-35
  Discriminator     /root/reference/hidden_models/discriminator.py:5-27
  EncoderDecoder    /root/reference/hidden_models/encoder_decoder.py:8-29
  train step        /root/reference/hidden_models/hidden.py:54-118

The reference's `options.HiDDenConfiguration` does not exist in its tree; the
field names below are the ones its modules read, the default values are
upstream HiDDeN's (SURVEY.md §8 header).
"""
import dataclasses

import numpy as np
import torch
import torch.nn as nn


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


class ConvBNRelu(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, padding=1), nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        return self.layers(x)


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.H, self.W = cfg.H, cfg.W
        c = cfg.encoder_channels
        self.conv_layers = nn.Sequential(*([ConvBNRelu(3, c)] + [ConvBNRelu(c, c) for _ in range(cfg.encoder_blocks - 1)]))
        self.after_concat_layer = ConvBNRelu(c + 3 + cfg.message_length, c)
        self.final_layer = nn.Conv2d(c, 3, kernel_size=1)

    def forward(self, image, message):
        B, L = message.shape
        m = message.view(B, L, 1, 1).expand(-1, -1, self.H, self.W)
        feat = self.conv_layers(image)
        # channel order: message, features, image (encoder.py:40)
        return self.final_layer(self.after_concat_layer(torch.cat([m, feat, image], dim=1)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.decoder_channels
        seq = [ConvBNRelu(3, c)] + [ConvBNRelu(c, c) for _ in range(cfg.decoder_blocks - 1)]
        seq += [ConvBNRelu(c, cfg.message_length), nn.AdaptiveAvgPool2d((1, 1))]
        self.layers = nn.Sequential(*seq)
        self.linear = nn.Linear(cfg.message_length, cfg.message_length)

    def forward(self, x):
        return self.linear(self.layers(x).flatten(1))


class Discriminator(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.discriminator_channels
        seq = [ConvBNRelu(3, c)] + [ConvBNRelu(c, c) for _ in range(cfg.discriminator_blocks - 1)]
        seq += [nn.AdaptiveAvgPool2d((1, 1))]
        self.before_linear = nn.Sequential(*seq)
        self.linear = nn.Linear(c, 1)

    def forward(self, x):
        return self.linear(self.before_linear(x).flatten(1))  # logits (discriminator.py:26)


class HiddenRef:
    """hidden.py:12-118 without VGG / tensorboard: nets, two Adam optimisers, one step."""

    def __init__(self, cfg, noiser):
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.discriminator = Discriminator(cfg)
        self.noiser = noiser  # callable image -> image
        self.opt_ed = torch.optim.Adam(list(self.encoder.parameters()) + list(self.decoder.parameters()))
        self.opt_d = torch.optim.Adam(self.discriminator.parameters())
        self.bce = nn.BCEWithLogitsLoss()
        self.mse = nn.MSELoss()

    def train_on_batch(self, images, messages):
        cfg = self.cfg
        B = images.shape[0]
        for m in (self.encoder, self.decoder, self.discriminator):
            m.train()
        self.opt_d.zero_grad()
        ones = torch.full((B, 1), 1.0)
        zeros = torch.full((B, 1), 0.0)
        l_dc = self.bce(self.discriminator(images), ones)
        l_dc.backward()
        encoded = self.encoder(images, messages)
        noised = self.noiser(encoded)
        decoded = self.decoder(noised)
        l_de = self.bce(self.discriminator(encoded.detach()), zeros)
        l_de.backward()
        grads_d = {n: p.grad.clone() for n, p in self.discriminator.named_parameters()}
        self.opt_d.step()
        self.opt_ed.zero_grad()
        l_adv = self.bce(self.discriminator(encoded), ones)
        l_enc = self.mse(encoded, images)
        l_dec = self.mse(decoded, messages)
        g = cfg.adversarial_loss * l_adv + cfg.encoder_loss * l_enc + cfg.decoder_loss * l_dec
        g.backward()
        grads = {
            "D": grads_d,
            "E": {n: p.grad.clone() for n, p in self.encoder.named_parameters()},
            "Dec": {n: p.grad.clone() for n, p in self.decoder.named_parameters()},
        }
        self.opt_ed.step()
        rounded = decoded.detach().numpy().round().clip(0, 1)
        biterr = float(np.sum(np.abs(rounded - messages.numpy())) / (B * messages.shape[1]))
        losses = {
            "loss           ": g.item(), "encoder_mse    ": l_enc.item(), "dec_mse        ": l_dec.item(),
            "bitwise-error  ": biterr, "adversarial_bce": l_adv.item(),
            "discr_cover_bce": l_dc.item(), "discr_encod_bce": l_de.item(),
        }
        return losses, (encoded, noised, decoded), grads
