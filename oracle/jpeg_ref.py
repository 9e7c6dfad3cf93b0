"""Oracle (test infrastructure): block-JPEG attack layers of the reference.

Follows /root/reference/noise_layers/jpeg.py:
  tables + quantisation      std_quantization          :52-82
  de-quantisation            std_reverse_quantization  :84-113
  8x8 DCT / IDCT             dct / idct                :115-145
  colour transforms          rgb2yuv / yuv2rgb         :147-163
  x255, pad to /8            yuv_dct                   :165-187
  un-pad, /255               idct_rgb                  :189-200
  4:2:0 by replication       subsampling               :202-211
  Jpeg / JpegSS / JpegMask   forward                   :226-240 / :259-273 / :295-306

Written with plain reshape + einsum on [B,3,H/8,8,W/8,8] blocks instead of the
reference's split/cat reshuffle; identical for every shape where that reshuffle
is self-consistent (padded H == padded W).  For padded H != padded W the reference
RAISES (jpeg.py:123-127 reuses one `split_num` for both axes; run on 16x32, 24x40 and
100x200 inputs it fails in its cat/chunk): there the natural block semantics used here
are an EXTENSION of this build, not parity with a reference result.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LUM = [
    [16, 11, 10, 16, 24, 40, 51, 61],
    [12, 12, 14, 19, 26, 58, 60, 55],
    [14, 13, 16, 24, 40, 57, 69, 56],
    [14, 17, 22, 29, 51, 87, 80, 62],
    [18, 22, 37, 56, 68, 109, 103, 77],
    [24, 35, 55, 64, 81, 104, 113, 92],
    [49, 64, 78, 87, 103, 121, 120, 101],
    [72, 92, 95, 98, 112, 100, 103, 99],
]
CHROMA = [
    [17, 18, 24, 47, 99, 99, 99, 99],
    [18, 21, 26, 66, 99, 99, 99, 99],
    [24, 26, 56, 99, 99, 99, 99, 99],
    [47, 66, 99, 99, 99, 99, 99, 99],
    [99, 99, 99, 99, 99, 99, 99, 99],
    [99, 99, 99, 99, 99, 99, 99, 99],
    [99, 99, 99, 99, 99, 99, 99, 99],
    [99, 99, 99, 99, 99, 99, 99, 99],
]


def scale_factor(Q):
    """jpeg.py:221"""
    return 2 - Q * 0.02 if Q >= 50 else 50 / Q


def quant_tables(scale):
    """jpeg.py:54-76 -- (table * scale).round().clamp(min=1), fp32, half-to-even."""
    lum = (torch.tensor(LUM, dtype=torch.float) * scale).round().clamp(min=1)
    chroma = (torch.tensor(CHROMA, dtype=torch.float) * scale).round().clamp(min=1)
    return lum, chroma


def dct_matrix():
    """jpeg.py:117-121 -- orthonormal DCT-II basis, float64 values stored as fp32."""
    coff = torch.zeros((8, 8), dtype=torch.float)
    coff[0, :] = 1 * np.sqrt(1 / 8)
    for i in range(1, 8):
        for j in range(8):
            coff[i, j] = np.cos(np.pi * i * (2 * j + 1) / (2 * 8)) * np.sqrt(2 / 8)
    return coff


def rgb2yuv(x):
    """jpeg.py:147-155 (no +128 offset on U,V)."""
    r, g, b = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    u = -0.1687 * r - 0.3313 * g + 0.5 * b
    v = 0.5 * r - 0.4187 * g - 0.0813 * b
    return torch.cat([y, u, v], 1)


def yuv2rgb(x):
    """jpeg.py:157-163"""
    y, u, v = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    r = y + 1.40198758 * v
    g = y - 0.344113281 * u - 0.714103821 * v
    b = y + 1.77197812 * u
    return torch.cat([r, g, b], 1)


def _blocks(x):
    B, C, H, W = x.shape
    return x.reshape(B, C, H // 8, 8, W // 8, 8).permute(0, 1, 2, 4, 3, 5)  # [B,C,nh,nw,8,8]


def _unblocks(b):
    B, C, nh, nw, _, _ = b.shape
    return b.permute(0, 1, 2, 4, 3, 5).reshape(B, C, nh * 8, nw * 8)


def subsampling(x, subsample):
    """jpeg.py:202-211 -- for subsample==2 odd rows of U,V take the even row above,
    then odd columns take the even column to the left (inside each 8x8 block, which
    for even offsets is the same as globally)."""
    if subsample != 2:
        return x
    y = x[:, 0:1]
    uv = x[:, 1:3]
    uv = uv[:, :, 0::2, :].repeat_interleave(2, dim=2)
    uv = uv[:, :, :, 0::2].repeat_interleave(2, dim=3)
    return torch.cat([y, uv], 1)


def round_ss(x):
    """jpeg.py:255-257; the condition is a detached float mask."""
    cond = (torch.abs(x) < 0.5).float()
    return cond * (x ** 3) + (1 - cond) * x


def mask_tables():
    """jpeg.py:288-291 -- keep Y[0:5,0:5], U/V[0:3,0:3]."""
    m = torch.zeros(3, 8, 8)
    m[0, :5, :5] = 1
    m[1:, :3, :3] = 1
    return m


def jpeg_layer(x, Q, mode="round", subsample=0):
    """x [B,3,H,W] (any range; the reference feeds [0,1]) -> same shape.
    mode: 'round' (Jpeg), 'ss' (JpegSS), 'mask' (JpegMask)."""
    B, C, H, W = x.shape
    assert C == 3
    coff = dct_matrix().to(x.dtype)
    img = x * 255
    ph = (8 - H % 8) % 8
    pw = (8 - W % 8) % 8
    img = F.pad(img, (0, pw, 0, ph))
    yuv = subsampling(rgb2yuv(img), subsample)
    blk = _blocks(yuv)
    dct = torch.matmul(torch.matmul(coff, blk), coff.t())
    if mode == "mask":
        deq = dct * mask_tables().to(x.dtype)[None, :, None, None]
    else:
        lum, chroma = quant_tables(scale_factor(Q))
        tbl = torch.stack([lum, chroma, chroma]).to(x.dtype)[None, :, None, None]
        q = dct / tbl
        q = torch.round(q) if mode == "round" else round_ss(q)
        deq = q * tbl
    rec = torch.matmul(torch.matmul(coff.t(), deq), coff)
    rgb = yuv2rgb(_unblocks(rec))
    rgb = rgb[:, :, :H, :W]
    return rgb / 255


def yuv_dct(x):
    """jpeg.py:165-187 with subsample 0: the DCT coefficients in image layout."""
    B, C, H, W = x.shape
    coff = dct_matrix()
    img = F.pad(x * 255, (0, (8 - W % 8) % 8, 0, (8 - H % 8) % 8))
    blk = _blocks(rgb2yuv(img))
    return _unblocks(torch.matmul(torch.matmul(coff, blk), coff.t()))


def layer_name(kind, Q):
    """jpeg.py:217,246,279"""
    return {"round": "Jpeg", "ss": "JpegSS", "mask": "JpegMask"}[kind] + str(Q)
