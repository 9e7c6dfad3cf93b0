"""models/conditional_jpeg_generator.py of the reference, the part SURVEY 8f row 1 names: the `conv()` block factory (:40-79),
`ResBlock` (:83-96), `QFAttention` (:185-200), `FBCNN` (:202-374), `QF_predictor` (:697-826) and `symm_pad` (:865-885), on the
HIP layer toolkit (glayers.py).  Constructors, forward contracts (NCHW f32 in / out) and state_dict keys are the reference's."""
import math

import torch
import torch.nn as nn

from .. import glayers as G
from .. import ops


def sequential(*args):
    """conditional_jpeg_generator.py:16-35: one module is returned as itself, nested Sequentials are flattened"""
    if len(args) == 1:
        return args[0]
    modules = []
    for m in args:
        if isinstance(m, nn.Sequential):
            modules.extend(m.children())
        elif isinstance(m, nn.Module):
            modules.append(m)
    return G.FusedSequential(*modules)      # (same children and keys; conv + activation pairs run as one autograd node)


_ACT = {"R": "relu", "r": "relu", "L": "lrelu", "l": "lrelu"}


def conv(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, mode="CBR", negative_slope=0.2):
    """conditional_jpeg_generator.py:40-79 for the letters FBCNN / QF_predictor use: C (Conv2d), T (ConvTranspose2d), R/r (ReLU),
    L/l (LeakyReLU(0.2))."""
    L = []
    for t in mode:
        if t == "C":
            L.append(G.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias))
        elif t == "T":
            L.append(G.ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding, bias))
        elif t in _ACT:
            if t in "Ll" and negative_slope != 0.2:
                raise NotImplementedError("LeakyReLU slope other than 0.2")
            L.append(G.Act(_ACT[t]))
        else:
            raise NotImplementedError(f"conv(): block letter '{t}' has no HIP layer (C, T, R, r, L, l are built)")
    return sequential(*L)


class ResBlock(nn.Module):
    """x + conv(relu(conv(x)))   (conditional_jpeg_generator.py:83-96)"""

    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, mode="CRC", negative_slope=0.2):
        super().__init__()
        assert in_channels == out_channels, "Only support in_channels==out_channels."
        if mode[0] in ["R", "L"]:
            mode = mode[0].lower() + mode[1:]
        self.res = conv(in_channels, out_channels, kernel_size, stride, padding, bias, mode, negative_slope)

    def forward(self, x):
        return G.add(x, self.res(x))


class QFAttention(nn.Module):
    """x + gamma * res(x) + beta   (conditional_jpeg_generator.py:185-200); gamma / beta [B,C]"""

    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1, bias=True, mode="CRC", negative_slope=0.2):
        super().__init__()
        assert in_channels == out_channels, "Only support in_channels==out_channels."
        if mode[0] in ["R", "L"]:
            mode = mode[0].lower() + mode[1:]
        self.res = conv(in_channels, out_channels, kernel_size, stride, padding, bias, mode, negative_slope)

    def forward(self, x, gamma, beta):
        return G.qf_attention(x, self.res(x), gamma, beta)


def upsample_convtranspose(in_channels=64, out_channels=3, kernel_size=2, stride=2, padding=0, bias=True, mode="2R", negative_slope=0.2):
    """conditional_jpeg_generator.py:123-129"""
    assert len(mode) < 4 and mode[0] in ["2", "3", "4"]
    k = int(mode[0])
    if k != 2:
        raise NotImplementedError("the HIP transposed conv is built for stride 2")
    return conv(in_channels, out_channels, k, k, padding, bias, mode.replace(mode[0], "T"), negative_slope)


def downsample_strideconv(in_channels=64, out_channels=64, kernel_size=2, stride=2, padding=0, bias=True, mode="2R", negative_slope=0.2):
    """conditional_jpeg_generator.py:147-153"""
    assert len(mode) < 4 and mode[0] in ["2", "3", "4"]
    k = int(mode[0])
    if k != 2:
        raise NotImplementedError("the HIP strided conv is built for stride 2")
    return conv(in_channels, out_channels, k, k, padding, bias, mode.replace(mode[0], "C"), negative_slope)


def _blocks(downsample_mode, upsample_mode):
    if downsample_mode != "strideconv":
        raise NotImplementedError(f"downsample mode [{downsample_mode}]: only strideconv (the reference's default) is built")
    if upsample_mode != "convtranspose":
        raise NotImplementedError(f"upsample mode [{upsample_mode}]: only convtranspose (the reference's default) is built")
    return downsample_strideconv, upsample_convtranspose


class FBCNN(nn.Module):
    """conditional_jpeg_generator.py:202-374: U-shaped restorer whose decoder ResBlocks are modulated by (gamma, beta) embedded from
    the quality factor.  forward(x [B,in_nc,H,W], qf_input [B,1]) -> (restored [B,out_nc,H,W], (x_m_1, x_m_2, x_m_3, x_m_4))."""

    def __init__(self, in_nc=3, out_nc=3, nc=[32, 64, 128, 256], nb=4, act_mode="R", downsample_mode="strideconv", qf_classes=6,
                 upsample_mode="convtranspose", dtype=torch.float32):
        super().__init__()
        down, up = _blocks(downsample_mode, upsample_mode)
        self.in_nc, self.out_nc, self.nb, self.nc, self.dtype = in_nc, out_nc, nb, nc, dtype
        m = "C" + act_mode + "C"
        self.m_head = conv(in_nc, nc[0], bias=True, mode="C")
        self.m_down1 = nn.ModuleList([down(nc[0], nc[1], bias=True, mode="2"), *[ResBlock(nc[1], nc[1], bias=True, mode=m) for _ in range(nb)]])
        self.m_down2 = nn.ModuleList([down(nc[1], nc[2], bias=True, mode="2"), *[ResBlock(nc[2], nc[2], bias=True, mode=m) for _ in range(nb)]])
        self.m_down3 = nn.ModuleList([down(nc[2], nc[2], bias=True, mode="2"), *[ResBlock(nc[2], nc[2], bias=True, mode=m) for _ in range(nb)]])
        self.m_body_encoder = sequential(*[ResBlock(nc[2], nc[2], bias=True, mode=m) for _ in range(nb)])
        self.m_up3 = nn.ModuleList([up(nc[2], nc[2], bias=True, mode="2"), *[QFAttention(nc[2], nc[2], bias=True, mode=m) for _ in range(nb)]])
        self.m_up2 = nn.ModuleList([up(nc[2], nc[1], bias=True, mode="2"), *[QFAttention(nc[1], nc[1], bias=True, mode=m) for _ in range(nb)]])
        self.m_up1 = nn.ModuleList([up(nc[1], nc[0], bias=True, mode="2"), *[QFAttention(nc[0], nc[0], bias=True, mode=m) for _ in range(nb)]])
        self.m_tail = conv(nc[0], out_nc, bias=True, mode="C")
        dim = 32
        # built and saved like the reference's, and like there not on the forward path (:292-306, :370-371)
        self.qf_downsample = sequential(G.Conv2d(3, dim, 4, 2, 1), G.Act("elu"), G.Conv2d(dim, dim * 2, 4, 2, 1), G.Act("elu"),
                                        G.Conv2d(dim * 2, dim * 4, 4, 2, 1), G.Act("elu"), G.Conv2d(dim * 4, dim * 8, 4, 2, 1), G.Act("elu"),
                                        G.Conv2d(dim * 8, dim * 16, 4, 2, 1))
        self.qf_embed = sequential(G.Linear(1, 512), G.Act("gelu"), G.Linear(512, 512), G.Act("gelu"), G.Linear(512, 512), G.Act("gelu"))
        self.to_gamma_3 = sequential(G.Linear(512, nc[2]), G.Act("sigmoid"))
        self.to_beta_3 = sequential(G.Linear(512, nc[2]), G.Act("tanh"))
        self.to_gamma_2 = sequential(G.Linear(512, nc[1]), G.Act("sigmoid"))
        self.to_beta_2 = sequential(G.Linear(512, nc[1]), G.Act("tanh"))
        self.to_gamma_1 = sequential(G.Linear(512, nc[0]), G.Act("sigmoid"))
        self.to_beta_1 = sequential(G.Linear(512, nc[0]), G.Act("tanh"))

    def forward(self, x, qf_input=None):
        if qf_input is None:
            raise ValueError("FBCNN.forward needs qf_input [B,1] (the reference has no fallback either: its qf_pred branch is commented out)")
        emb = self.qf_embed(G.vector_in(qf_input.reshape(x.shape[0], -1)))      # the embedding stack stays f32
        gamma_3, beta_3 = self.to_gamma_3(emb), self.to_beta_3(emb)
        gamma_2, beta_2 = self.to_gamma_2(emb), self.to_beta_2(emb)
        gamma_1, beta_1 = self.to_gamma_1(emb), self.to_beta_1(emb)

        h, w = x.shape[-2:]
        pad_b, pad_r = int(math.ceil(h / 8) * 8 - h), int(math.ceil(w / 8) * 8 - w)
        x = G.to_nhwc(x, self.dtype, (0, pad_r, 0, pad_b), ops.PAD_REPLICATE)     # nn.ReplicationPad2d((0, r, 0, b)) fused into the layout change

        x1 = self.m_head(x)
        x2 = self.m_down1[0](x1)
        for i in range(self.nb):
            x2 = self.m_down1[i + 1](x2)
        x3 = self.m_down2[0](x2)
        for i in range(self.nb):
            x3 = self.m_down2[i + 1](x3)
        x4 = self.m_down3[0](x3)
        for i in range(self.nb):
            x4 = self.m_down3[i + 1](x4)
        x_m_1 = self.m_body_encoder(x4)
        x = G.add(x_m_1, x4)
        x_m_2 = self.m_up3[0](x)
        for i in range(self.nb):
            x_m_2 = self.m_up3[i + 1](x_m_2, gamma_3, beta_3)
        x = G.add(x_m_2, x3)
        x_m_3 = self.m_up2[0](x)
        for i in range(self.nb):
            x_m_3 = self.m_up2[i + 1](x_m_3, gamma_2, beta_2)
        x = G.add(x_m_3, x2)
        x_m_4 = self.m_up1[0](x)
        for i in range(self.nb):
            x_m_4 = self.m_up1[i + 1](x_m_4, gamma_1, beta_1)
        x = G.add(x_m_4, x1)
        x = self.m_tail(x)
        out = G.to_nchw(x, self.out_nc, h, w)                                     # x[..., :h, :w]
        nc = self.nc
        return out, (G.to_nchw(x_m_1, nc[2]), G.to_nchw(x_m_2, nc[2]), G.to_nchw(x_m_3, nc[1]), G.to_nchw(x_m_4, nc[0]))


def symm_pad(im, padding, dtype=torch.float32):
    """conditional_jpeg_generator.py:865-885 on an NCHW f32 image; returns NCHW f32 (the networks fuse the padding into their layout
    change instead of calling this)."""
    left, right, top, bottom = padding
    x = G.to_nhwc(im, dtype, (left, right, top, bottom), ops.PAD_SYMMETRIC)
    return G.to_nchw(x, im.shape[1])


class _BicubicTo(torch.autograd.Function):
    """F.interpolate(x, size=[oh, ow], mode='bicubic') on f32 planes (the attack set's ATen-exact resample kernels, ops.resample_*)"""

    @staticmethod
    def forward(ctx, x, oh, ow):
        x = x.float().contiguous()
        ctx.dims = (x.shape[2], x.shape[3])
        return ops.resample_fwd(x, (0, x.shape[2], 0, x.shape[3]), (oh, ow), ops.BICUBIC)

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.dims
        return ops.resample_bwd(g.float().contiguous(), None, (H, W), (0, H, 0, W), ops.BICUBIC), None, None


class QF_predictor(nn.Module):
    """conditional_jpeg_generator.py:697-826: constrained 5x5 Bayar conv on the symmetrically padded image, three
    (nb ResBlocks, stride-2 conv) stages to 192 channels, nb ResBlocks, then nb ResBlocks + global pool + 3 Linear.
    forward(x [B,3,H,W]) -> (conv_bayar [B,3,H,W], qf [B,classes]).
    crop_pred=True (:772-784, :817-821): the head is the pool + 3 Linear alone, and the first output is `to_img` (1x1, 192 -> 1, no bias)
    of the encoder's features resized to 512 x 512 (bicubic): (img [B,1,512,512], qf)."""

    def __init__(self, in_nc=3, out_nc=3, nc=[32, 64, 128, 256], nb=4, act_mode="R", downsample_mode="strideconv", classes=5, crop_pred=False,
                 upsample_mode="convtranspose", dtype=torch.float32):
        super().__init__()
        down, _ = _blocks(downsample_mode, upsample_mode)
        self.in_nc, self.nb, self.nc, self.crop_pred, self.classes, self.dtype = in_nc, nb, nc, crop_pred, classes, dtype
        m = "C" + act_mode + "C"
        self.BayarConv2D = G.Conv2d(3, 3, 5, 1, 0, bias=False)
        self.relu = G.Act("lrelu")
        self.m_head_A = conv(3, nc[0], bias=True, mode="C")
        self.m_down1_A = sequential(*[ResBlock(nc[0], nc[0], bias=True, mode=m) for _ in range(nb)], down(nc[0], nc[1], bias=True, mode="2"))
        self.m_down2_A = sequential(*[ResBlock(nc[1], nc[1], bias=True, mode=m) for _ in range(nb)], down(nc[1], nc[2], bias=True, mode="2"))
        self.m_down3_A = sequential(*[ResBlock(nc[2], nc[2], bias=True, mode=m) for _ in range(nb)], down(nc[2], 192, bias=True, mode="2"))
        self.m_body_encoder_A = sequential(*[ResBlock(192, 192, bias=True, mode=m) for _ in range(nb)])
        head = [G.GlobalAvgPool(), G.Flatten(), G.Linear(192, 192), G.Act("gelu"), G.Linear(192, 192), G.Act("gelu"), G.Linear(192, classes)]
        if crop_pred:
            self.to_img = G.Conv2d(192, 1, 1, 1, 0, bias=False)
            self.qf_pred = sequential(*head)
        else:
            self.qf_pred = sequential(*[ResBlock(192, 192, bias=True, mode=m) for _ in range(nb)], *head)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"QF_predictor expects [B,3,H,W], got {tuple(x.shape)}")
        ops.bayar_constrain_(self.BayarConv2D.weight.data)                        # :814-817, in place on the parameter like the reference
        xp = G.to_nhwc(x, self.dtype, (2, 2, 2, 2), ops.PAD_SYMMETRIC)            # symm_pad(x, (2,2,2,2))
        e0 = self.BayarConv2D(xp)
        x1 = self.m_head_A(e0)
        x2 = self.m_down1_A(x1)
        x3 = self.m_down2_A(x2)
        x4 = self.m_down3_A(x3)
        x_pred = self.m_body_encoder_A(x4)
        qf = self.qf_pred(x_pred)
        if self.crop_pred:
            img = _BicubicTo.apply(G.to_nchw(self.to_img(x_pred), 1), 512, 512)
            return img, G.vector_out(qf, self.classes)
        return G.to_nchw(e0, 3), G.vector_out(qf, self.classes)
