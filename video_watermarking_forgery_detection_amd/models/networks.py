"""models/networks.py of the reference, the part SURVEY 8f row 1 names: `Discriminator` (networks.py:631-749) on the HIP layer
toolkit (glayers.py).  Same constructor, same forward contract (NCHW f32 image in, [B,1,H/32,W/32] f32 out), same state_dict
keys (`init_conv.0.weight_orig`, `.weight_u`, `.weight_v`, ..., `conv5.0.weight`)."""
import torch
import torch.nn as nn

from .. import glayers as G


def spectral_norm_conv(cin, cout, k, stride, padding, use_spectral_norm):
    # networks.py:1381-1385 spectral_norm(nn.Conv2d(.., bias=not use_spectral_norm), use_spectral_norm)
    cls = G.SpectralNormConv2d if use_spectral_norm else G.Conv2d
    return cls(cin, cout, k, stride, padding, bias=not use_spectral_norm)


class BaseNetwork(nn.Module):
    def init_weights(self, init_type="kaiming", gain=0.02):
        """networks.py:104-129.  The reference initialises `m.weight.data`; under spectral norm that is the derived attribute the
        next forward overwrites, so `weight_orig` keeps nn.Conv2d's default initialisation -- as here."""

        def init_func(m):
            if isinstance(m, G.SpectralNormConv2d) or not isinstance(m, (G.Conv2d, G.Linear, G.ConvTranspose2d)):
                return
            w = m.weight.data
            if init_type == "normal":
                nn.init.normal_(w, 0.0, gain)
            elif init_type == "xavier":
                nn.init.xavier_normal_(w, gain=gain)
            elif init_type == "kaiming":
                nn.init.kaiming_normal_(w, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                nn.init.orthogonal_(w, gain=gain)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)

        self.apply(init_func)


class Discriminator(BaseNetwork):
    """networks.py:631-749: five (4x4 stride-2 conv, GELU, 3x3 conv, GELU) stages 3 -> 32 -> 64 -> 128 -> 256 -> 512 under spectral
    norm, a 1x1 conv to one channel, sigmoid.  `in_channels` and `use_SRM` are accepted and, as in the reference, unused (the first
    conv is hard-wired to 3 channels)."""

    def __init__(self, in_channels, use_sigmoid=True, use_spectral_norm=True, init_weights=True, use_SRM=False, dtype=torch.float32):
        super().__init__()
        self.use_sigmoid = use_sigmoid
        self.use_SRM = use_SRM
        self.in_channels = in_channels
        self.dtype = dtype
        dim = 32
        sn = use_spectral_norm

        def stage(cin, cout):
            return G.FusedSequential(spectral_norm_conv(cin, cout, 4, 2, 1, sn), G.Act("gelu"), spectral_norm_conv(cout, cout, 3, 1, 1, sn), G.Act("gelu"))

        self.init_conv = stage(3, dim)
        self.conv1 = stage(dim, dim * 2)
        self.conv2 = stage(dim * 2, dim * 4)
        self.conv3 = stage(dim * 4, dim * 8)
        self.conv4 = stage(dim * 8, dim * 16)
        self.conv5 = nn.Sequential(G.Conv2d(dim * 16, 1, 1, 1, 0, bias=False))
        self._sigmoid = G.Act("sigmoid")
        if init_weights:
            self.init_weights()

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"Discriminator expects [B,3,H,W], got {tuple(x.shape)}")
        h = G.to_nhwc(x, self.dtype)
        for blk in (self.init_conv, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            h = blk(h)
        if self.use_sigmoid:
            h = self._sigmoid(h)
        return G.to_nchw(h, 1)
