"""Oracle (test infrastructure): the tamper-localisation UNet.

Follows /root/reference/network/UNet.py:7-97 -- 4-level UNet, blocks of
2 x (Conv3x3 no-bias + BatchNorm + ReLU), MaxPool2 down, ConvTranspose2d k2s2 up,
skip concat (upsampled first, skip second), 1x1 conv + sigmoid head.  Same
state_dict keys (`encoder1.enc1conv1.weight`, `enc1norm1.*`, `upconv4.*`, `conv.*`).
"""
from collections import OrderedDict

import torch
import torch.nn as nn


def _block(cin, cout, name):
    od = OrderedDict()
    for i, ci in ((1, cin), (2, cout)):
        od[f"{name}conv{i}"] = nn.Conv2d(ci, cout, 3, padding=1, bias=False)
        od[f"{name}norm{i}"] = nn.BatchNorm2d(cout)
        od[f"{name}relu{i}"] = nn.ReLU()
    return nn.Sequential(od)


class UNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=1, init_features=32):
        super().__init__()
        f = init_features
        chans = [f, 2 * f, 4 * f, 8 * f]
        prev = in_channels
        for lvl, c in enumerate(chans, 1):
            setattr(self, f"encoder{lvl}", _block(prev, c, f"enc{lvl}"))
            setattr(self, f"pool{lvl}", nn.MaxPool2d(2, 2))
            prev = c
        self.bottleneck = _block(8 * f, 16 * f, "bottleneck")
        prev = 16 * f
        for lvl in (4, 3, 2, 1):
            c = chans[lvl - 1]
            setattr(self, f"upconv{lvl}", nn.ConvTranspose2d(prev, c, 2, 2))
            setattr(self, f"decoder{lvl}", _block(2 * c, c, f"dec{lvl}"))
            prev = c
        self.conv = nn.Conv2d(f, out_channels, 1)

    def forward(self, x):
        skips = []
        for lvl in (1, 2, 3, 4):
            x = getattr(self, f"encoder{lvl}")(x)
            skips.append(x)
            x = getattr(self, f"pool{lvl}")(x)
        x = self.bottleneck(x)
        for lvl in (4, 3, 2, 1):
            x = getattr(self, f"upconv{lvl}")(x)
            x = getattr(self, f"decoder{lvl}")(torch.cat((x, skips[lvl - 1]), dim=1))
        return torch.sigmoid(self.conv(x))
