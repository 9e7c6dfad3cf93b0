"""Oracle (test infrastructure): stencil / resample attacks + Quantization.

  gaussian_kernel / gaussian_blur   /root/reference/noise_layers/gaussian_blur.py:17-56
  median_blur                       /root/reference/noise_layers/middle_filter.py:5-13
                                    (kornia.filters.MedianBlur -- third party, version
                                    unpinned, ABSENT from this image: PARITY UNPINNED.
                                    Restated from kornia's published algorithm: zero-pad
                                    by k//2, gather the k*k window, torch.median over it.)
  resize                            /root/reference/noise_layers/resize.py:28-55
  crop                              /root/reference/noise_layers/crop.py:13-55
  quantization                      /root/reference/models/modules/Quantization.py:4-21
  combined                          /root/reference/noise_layers/combined.py:6-20
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F


def gaussian_kernel(kernel_size=3, sigma=2.0):
    """gaussian_blur.py:17-38 -- normalised 2-D Gaussian, fp32."""
    x_coord = torch.arange(kernel_size)
    x_grid = x_coord.repeat(kernel_size).view(kernel_size, kernel_size)
    y_grid = x_grid.t()
    xy = torch.stack([x_grid, y_grid], dim=-1).float()
    mean = (kernel_size - 1) / 2.
    var = sigma ** 2.
    k = (1. / (2. * math.pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2., dim=-1) / (2 * var))
    return k / torch.sum(k)


def gaussian_blur(x, kernel_size=3, sigma=2.0):
    """depthwise conv, zero padding (k-1)/2, no bias (gaussian_blur.py:44-56)."""
    C = x.shape[1]
    k = gaussian_kernel(kernel_size, sigma).to(x.dtype)
    w = k.view(1, 1, kernel_size, kernel_size).repeat(C, 1, 1, 1)
    return F.conv2d(x, w, padding=int((kernel_size - 1) / 2), groups=C)


def median_blur(x, k=3):
    """per-channel k x k median, zero padding, odd k.  Gradient goes to the selected
    element (torch.median semantics)."""
    B, C, H, W = x.shape
    p = k // 2
    cols = F.unfold(x.reshape(B * C, 1, H, W), kernel_size=k, padding=p)  # [BC, k*k, H*W]
    med = torch.median(cols, dim=1)[0]
    return med.view(B, C, H, W)


def resize(x, ratio, mode="bicubic"):
    """resize.py:38-53: to int(r*H) x int(r*W) and back, clamp [0,1]."""
    H, W = x.shape[2], x.shape[3]
    nh, nw = int(ratio * H), int(ratio * W)
    out = F.interpolate(x, size=[nh, nw], mode=mode)
    rec = F.interpolate(out, size=[H, W], mode=mode)
    return torch.clamp(rec, 0, 1)


def random_rectangle(shape, height_ratio, width_ratio):
    """crop.py:13-30 (numpy global RNG)."""
    ih, iw = shape[2], shape[3]
    rh, rw = int(height_ratio * ih), int(width_ratio * iw)
    hs = 0 if rh == ih else np.random.randint(0, ih - rh)
    ws = 0 if rw == iw else np.random.randint(0, iw - rw)
    return hs, hs + rh, ws, ws + rw


def crop(x, apex=None, min_rate=0.5, max_rate=1.0):
    """crop.py:32-55 -> (bilinear-resized crop, apex).  The two ratio draws happen
    even when apex is given (the RNG stream advances the same way)."""
    if min_rate:
        hr = min_rate + (max_rate - min_rate) * np.random.rand()
        wr = min_rate + (max_rate - min_rate) * np.random.rand()
    else:
        hr = 0.3 + 0.7 * np.random.rand()
        wr = 0.3 + 0.7 * np.random.rand()
    hr = min(hr, wr + 0.2)
    wr = min(wr, hr + 0.2)
    if apex is None:
        apex = random_rectangle(x.shape, hr, wr)
    hs, he, ws, we = apex
    sub = x[:, :, hs:he, ws:we]
    return F.interpolate(sub, size=[x.shape[2], x.shape[3]], mode="bilinear"), (hs, he, ws, we)


class _Quant(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return (x * 255.).round() / 255.

    @staticmethod
    def backward(ctx, g):
        return g


def quantization(x):
    return _Quant.apply(x)


def combined_pick(n, id=None):
    """combined.py:15-17 -- python `random.randint` unless 0 <= id < n."""
    if id is None or id >= n:
        id = random.randint(0, n - 1)
    return id
