"""models/invertible_net.py of the reference, the part SURVEY 8f row 2 names: the invertible watermark embedder
`Inveritible_Decolorization_PAMI` (:476-531) -- `HaarDownsampling` / `HaarUpsampling` (:178-292), `RNVPCouplingBlock` (:122-175) with
`ResBlock` (:326-366) or `DenseBlock` (:302-324) subnets -- on the HIP layer toolkit (glayers.py, csrc/inn.hip, csrc/gconv.hip).
Same constructors, the same forward(x, rev) contract on NCHW f32 tensors (rev=True returns (out, out_middle)), the same state_dict
keys (`operations_down.0.haar_weights`, `operations_down.1.s1.conv1.0.weight`, ...)."""
import torch
import torch.nn as nn

from .. import glayers as G


def _haar_weights(channels):
    # :187-199 -- kept as a frozen parameter so checkpoints round-trip; the kernel (csrc/inn.hip) has these signs built in
    w = torch.ones(4, 1, 2, 2)
    w[1, 0, 0, 1] = -1
    w[1, 0, 1, 1] = -1
    w[2, 0, 1, 0] = -1
    w[2, 0, 1, 1] = -1
    w[3, 0, 1, 0] = -1
    w[3, 0, 0, 1] = -1
    return torch.cat([w] * channels, 0)


def _xavier_(convs, scale):
    # initialize_weights_xavier (:27-44)
    for m in convs:
        nn.init.xavier_normal_(m.weight)
        m.weight.data *= scale
        if m.bias is not None:
            m.bias.data.zero_()


def _kaiming_(convs, scale):
    # initialize_weights (:7-24)
    for m in convs:
        nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in")
        m.weight.data *= scale
        if m.bias is not None:
            m.bias.data.zero_()


class HaarDownsampling(nn.Module):
    """:178-247.  forward(x [B,H,W,cpad(C)]) -> [B,H/2,W/2,cpad(4C)], channel 4c+k = 0.5 * rebalance * Haar_k(channel c);
    rev: the synthesis scaled by 0.5 / rebalance.  order_by_wavelet (:207-218, the index permutation out[:, perm] / x[:, perm_inv]):
    wavelet k of channel c is channel k*C + c -- the kernels write / read that order directly, no gather."""

    def __init__(self, dims_in, order_by_wavelet=False, rebalance=1.0):
        super().__init__()
        self.permute = bool(order_by_wavelet)
        self.in_channels = dims_in[0][0]
        self.fac_fwd = 0.5 * rebalance
        self.fac_rev = 0.5 / rebalance
        self.haar_weights = nn.Parameter(_haar_weights(self.in_channels), requires_grad=False)

    def forward(self, x, rev=False):
        if not rev:
            return G.haar_down(x, self.in_channels, self.fac_fwd, self.permute)
        return G.haar_up(x, self.in_channels, self.fac_rev, self.permute)


class HaarUpsampling(nn.Module):
    """:250-292.  forward: [B,H,W,cpad(4C)] -> [B,2H,2W,cpad(C)] with the 0.5-scaled filters; rev: the analysis with the same."""

    def __init__(self, dims_in):
        super().__init__()
        self.in_channels = dims_in[0][0] // 4
        self.haar_weights = nn.Parameter(_haar_weights(self.in_channels) * 0.5, requires_grad=False)

    def forward(self, x, rev=False):
        if rev:
            return G.haar_down(x, self.in_channels, 0.5)
        return G.haar_up(x, self.in_channels, 0.5)


class ResBlock(nn.Module):
    """the coupling subnet (:326-366): four (3x3 conv to 64, ELU), then a 3x3 conv of cat(x, features) to channel_out"""

    def __init__(self, channel_in, channel_out, use_spectral_norm=False, init="xavier"):
        super().__init__()
        feature = 64
        cls = G.SpectralNormConv2d if use_spectral_norm else G.Conv2d
        self.channel_in, self.channel_out, self.feature = channel_in, channel_out, feature
        self.conv1 = G.ConvAct(cls(channel_in, feature, 3, 1, 1), "elu")
        self.conv2 = G.ConvAct(cls(feature, feature, 3, 1, 1), "elu")
        self.conv3 = G.ConvAct(cls(feature, feature, 3, 1, 1), "elu")
        self.conv4 = G.ConvAct(cls(feature, feature, 3, 1, 1), "elu")
        self.conv5 = G.Conv2d(feature + channel_in, channel_out, 3, 1, 1)
        if not use_spectral_norm:
            (_xavier_ if init == "xavier" else _kaiming_)([self.conv1[0], self.conv2[0], self.conv3[0], self.conv4[0]], 0.1)
        _kaiming_([self.conv5], 0)

    def forward(self, x):
        r = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        return self.conv5(G.chan_cat(x, self.channel_in, r, self.feature))


class DenseBlock(nn.Module):
    """:302-324: five 3x3 convs over the growing concatenation (x, x1, ..), ELU between"""

    def __init__(self, channel_in, channel_out, init="xavier", gc=32, bias=True):
        super().__init__()
        self.channel_in, self.gc = channel_in, gc
        self.conv1 = G.Conv2d(channel_in, gc, 3, 1, 1, bias=bias)
        self.conv2 = G.Conv2d(channel_in + gc, gc, 3, 1, 1, bias=bias)
        self.conv3 = G.Conv2d(channel_in + 2 * gc, gc, 3, 1, 1, bias=bias)
        self.conv4 = G.Conv2d(channel_in + 3 * gc, gc, 3, 1, 1, bias=bias)
        self.conv5 = G.Conv2d(channel_in + 4 * gc, channel_out, 3, 1, 1, bias=bias)
        self.lrelu = G.Act("elu")
        (_xavier_ if init == "xavier" else _kaiming_)([self.conv1, self.conv2, self.conv3, self.conv4], 0.1)
        _kaiming_([self.conv5], 0)

    def forward(self, x):
        cat, n = x, self.channel_in
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            cat, n = G.chan_cat(cat, n, self.lrelu(conv(cat)), self.gc), n + self.gc
        return self.conv5(cat)


PARALLEL_SUBNETS = True      # the s / t subnets of a coupling side by side on two streams (see _pair); False = one stream, the same results bit for bit
_SIDE_STREAMS = {}


def _pair(f, g, x):
    """(f(x), g(x)).  With PARALLEL_SUBNETS on a GPU, f runs on a side stream forked from the caller's and joined before the pair is
    used: the two subnets of a coupling share their input and nothing else, and at the embedder's sizes (8k-131k pixels per layer) a
    launch fills 32-256 workgroups for ~10 us of fixed cost -- two of them side by side cost little more than one.  Autograd runs every
    backward node on its forward's stream and synchronises the edges between streams itself, so the backward forks the same way.  Every
    tensor crossing streams is recorded on its reader's stream (the allocator re-issues a block only after that stream has passed);
    the kernels are the same launches and autograd sums every gradient in the same order: the result is the one-stream one bit for bit
    (tests/test_gpu_f2.py; tools/inn_sanity_modes.py over 10 optimiser steps, eager and replayed).  The FIRST subnet is the one that goes
    to the side stream: a hipGraph replay runs the two branches side by side only when the forked branch is enqueued first (measured:
    104.7 -> 83.8 ms per replayed step that way, 108 ms with the second subnet forked instead -- profiles/r04_experiments.txt)."""
    if not (PARALLEL_SUBNETS and x.is_cuda):
        return f(x), g(x)
    main = torch.cuda.current_stream()
    side = _SIDE_STREAMS.get(x.device.index)
    if side is None:
        side = _SIDE_STREAMS[x.device.index] = torch.cuda.Stream(device=x.device)
    side.wait_stream(main)
    x.record_stream(side)
    with torch.cuda.stream(side):
        a = f(x)                  # (f's nodes before g's, as in the sequential order: autograd then sums the gradients of x in the same order)
    b = g(x)
    main.wait_stream(side)
    a.record_stream(main)
    return a, b


class RNVPCouplingBlock(nn.Module):
    """:122-175: y1 = e(s2(x2)) * x1 + t2(x2), y2 = e(s1(y1)) * x2 + t1(y1), e(s) = exp(clamp * (2 sigmoid(s) - 1)) + 1e-4; rev undoes it"""

    def __init__(self, dims_in, subnet_constructor=None, clamp=1.0):
        super().__init__()
        channels = dims_in[0][0]
        self.channels = channels
        self.split_len1 = channels // 2
        self.split_len2 = channels - channels // 2
        self.clamp = clamp
        self.affine_eps = 0.0001
        self.s1 = subnet_constructor(self.split_len1, self.split_len2)
        self.t1 = subnet_constructor(self.split_len1, self.split_len2)
        self.s2 = subnet_constructor(self.split_len2, self.split_len1)
        self.t2 = subnet_constructor(self.split_len2, self.split_len1)

    def forward(self, x, rev=False):
        n1, n2 = self.split_len1, self.split_len2
        x1, x2 = G.chan_slice(x, 0, n1), G.chan_slice(x, n1, n2)
        if not rev:
            y1 = G.coupling(x1, *_pair(self.s2, self.t2, x2), self.clamp, self.affine_eps, False)
            y2 = G.coupling(x2, *_pair(self.s1, self.t1, y1), self.clamp, self.affine_eps, False)
        else:
            y2 = G.coupling(x2, *_pair(self.s1, self.t1, x1), self.clamp, self.affine_eps, True)
            y1 = G.coupling(x1, *_pair(self.s2, self.t2, y2), self.clamp, self.affine_eps, True)
        return G.chan_cat(y1, n1, y2, n2)


class Inveritible_Decolorization_PAMI(nn.Module):
    """:476-531: down_num x (Haar analysis, block_num[i] couplings), then down_num x (Haar synthesis, couplings of the mirrored
    block_num with none at full resolution).  forward(x [B,C,H,W]) -> [B,C,H,W]; forward(x, rev=True) -> (recovered, out_middle),
    out_middle [B, C * 4**down_num, H / 2**down_num, W / 2**down_num] = the tensor between the two halves."""

    def __init__(self, dims_in=[[3, 64, 64]], down_num=3, block_num=[8, 8, 8], subnet_constructor=ResBlock, dtype=torch.float32):
        super().__init__()
        self.dtype = dtype
        self.channels_in = dims_in[0][0]
        self.down_num = down_num
        dims = [list(dims_in[0])]           # the reference mutates its argument in place (:484-486); a copy here
        ups, downs = [], []
        for i in range(down_num):
            downs.append(HaarDownsampling(dims))
            dims[0][0], dims[0][1], dims[0][2] = dims[0][0] * 4, dims[0][1] // 2, dims[0][2] // 2
            for _ in range(block_num[i]):
                downs.append(RNVPCouplingBlock(dims, subnet_constructor=subnet_constructor, clamp=1.0))
        self.channels_mid = dims[0][0]
        up_blocks = list(block_num[:-1][::-1]) + [0]
        for i in range(down_num):
            ups.append(HaarUpsampling(dims))
            dims[0][0], dims[0][1], dims[0][2] = dims[0][0] // 4, dims[0][1] * 2, dims[0][2] * 2
            for _ in range(up_blocks[i]):
                ups.append(RNVPCouplingBlock(dims, subnet_constructor=subnet_constructor, clamp=1.0))
        self.operations_up = nn.ModuleList(ups)
        self.operations_down = nn.ModuleList(downs)

    def forward(self, x, rev=False):
        m = 1 << self.down_num
        if x.dim() != 4 or x.shape[1] != self.channels_in or x.shape[2] % m or x.shape[3] % m:
            raise ValueError(f"expected [B,{self.channels_in},H,W] with H, W multiples of {m}, got {tuple(x.shape)}")
        out = G.to_nhwc(x, self.dtype)
        if not rev:
            for op in self.operations_down:
                out = op(out, False)
            for op in self.operations_up:
                out = op(out, False)
            return G.to_nchw(out, self.channels_in)
        for op in reversed(self.operations_up):
            out = op(out, True)
        middle = G.to_nchw(out, self.channels_mid)
        for op in reversed(self.operations_down):
            out = op(out, True)
        return G.to_nchw(out, self.channels_in), middle
