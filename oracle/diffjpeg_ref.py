"""Oracle (test infrastructure): the DiffJPEG pipeline of the reference.

Follows /root/reference/utils/JPEG.py:
  tables (transposed)          :96-108
  rgb_to_ycbcr_jpeg            :115-135   (+128 on Cb,Cr)
  chroma_subsampling 2x2 avg   :139-160
  block_splitting              :164-181
  dct_8x8 (x-128, tensordot)   :185-208
  y_quantize / c_quantize      :212-253
  compress_jpeg                :256-291
  y/c_dequantize               :295-328
  idct_8x8                     :332-354
  block_merging                :358-376
  chroma_upsampling (nearest)  :380-404
  ycbcr_to_rgb_jpeg            :408-428
  decompress_jpeg (+clamp)     :431-469
  diff_round / round_only_at_0 :472-484
  quality_to_factor            :487-498
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F

Y_TABLE = torch.from_numpy(np.array(
    [[16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55],
     [14, 13, 16, 24, 40, 57, 69, 56], [14, 17, 22, 29, 51, 87, 80, 62],
     [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
     [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]],
    dtype=np.float32).T.copy())
_c = np.empty((8, 8), dtype=np.float32)
_c.fill(99)
_c[:4, :4] = np.array([[17, 18, 24, 47], [18, 21, 26, 66], [24, 26, 56, 99], [47, 66, 99, 99]]).T
C_TABLE = torch.from_numpy(_c)


def quality_to_factor(quality):
    if quality < 50:
        quality = 5000. / quality
    else:
        quality = 200. - quality * 2
    return quality / 100.


def diff_round(x):
    return torch.round(x) + (x - torch.round(x)) ** 3


def round_only_at_0(x):
    cond = (torch.abs(x) < 0.5).float()
    return cond * (x ** 3) + (1 - cond) * x


def _dct_tensor():
    t = np.zeros((8, 8, 8, 8), dtype=np.float32)
    for x, y, u, v in itertools.product(range(8), repeat=4):
        t[x, y, u, v] = np.cos((2 * x + 1) * u * np.pi / 16) * np.cos((2 * y + 1) * v * np.pi / 16)
    alpha = np.array([1. / np.sqrt(2)] + [1] * 7)
    return torch.from_numpy(t).float(), torch.from_numpy(np.outer(alpha, alpha) * 0.25).float()


def _idct_tensor():
    t = np.zeros((8, 8, 8, 8), dtype=np.float32)
    for x, y, u, v in itertools.product(range(8), repeat=4):
        t[x, y, u, v] = np.cos((2 * u + 1) * x * np.pi / 16) * np.cos((2 * v + 1) * y * np.pi / 16)
    alpha = np.array([1. / np.sqrt(2)] + [1] * 7)
    return torch.from_numpy(t).float(), torch.from_numpy(np.outer(alpha, alpha)).float()


def _split(img):  # [B,H,W] -> [B, H*W/64, 8, 8]
    B, H, W = img.shape
    return img.view(B, H // 8, 8, -1, 8).permute(0, 1, 3, 2, 4).contiguous().view(B, -1, 8, 8)


def _merge(p, H, W):
    B = p.shape[0]
    return p.view(B, H // 8, W // 8, 8, 8).permute(0, 1, 3, 2, 4).contiguous().view(B, H, W)


def compress(x, factor, rounding):
    """x [B,3,H,W] in [0,1], H,W multiples of 16 -> (y [B,HW/64,8,8], cb, cr [B,HW/256,8,8])."""
    img = (x * 255).permute(0, 2, 3, 1)
    m = torch.from_numpy(np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5],
                                   [0.5, -0.418688, -0.081312]], dtype=np.float32).T.copy())
    ycc = torch.tensordot(img, m, dims=1) + torch.tensor([0., 128., 128.])
    ycc_c = ycc.permute(0, 3, 1, 2)
    cb = F.avg_pool2d(ycc_c[:, 1:2], 2, 2)[:, 0]
    cr = F.avg_pool2d(ycc_c[:, 2:3], 2, 2)[:, 0]
    yy = ycc[..., 0]
    T, scale = _dct_tensor()
    outs = []
    for comp, tbl in ((yy, Y_TABLE), (cb, C_TABLE), (cr, C_TABLE)):
        blk = _split(comp) - 128
        d = scale * torch.tensordot(blk, T, dims=2)
        outs.append(rounding(d / (tbl * factor)))
    return tuple(outs)


def decompress(y, cb, cr, H, W, factor):
    T, alpha = _idct_tensor()
    comps = []
    for comp, tbl, (h, w) in ((y, Y_TABLE, (H, W)), (cb, C_TABLE, (H // 2, W // 2)), (cr, C_TABLE, (H // 2, W // 2))):
        d = comp * (tbl * factor)
        img = 0.25 * torch.tensordot(d * alpha, T, dims=2) + 128
        comps.append(_merge(img, h, w))
    yy, cbb, crr = comps
    cbb = cbb.repeat_interleave(2, 1).repeat_interleave(2, 2)
    crr = crr.repeat_interleave(2, 1).repeat_interleave(2, 2)
    img = torch.stack([yy, cbb, crr], dim=3)
    m = torch.from_numpy(np.array([[1., 0., 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]], dtype=np.float32).T.copy())
    rgb = torch.tensordot(img + torch.tensor([0, -128., -128.]), m, dims=1).permute(0, 3, 1, 2)
    rgb = torch.min(255 * torch.ones_like(rgb), torch.max(torch.zeros_like(rgb), rgb))
    return rgb / 255


def diffjpeg(x, quality=75, rounding=round_only_at_0):
    """DiffJPEG.forward (:535-540) with the instance's quality/rounding."""
    B, C, H, W = x.shape
    f = quality_to_factor(quality)
    y, cb, cr = compress(x, f, rounding)
    return decompress(y, cb, cr, H, W, f)
