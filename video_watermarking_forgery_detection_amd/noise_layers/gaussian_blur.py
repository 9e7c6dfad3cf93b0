"""GaussianBlur -- mirror of the reference's noise_layers/gaussian_blur.py:7-56: depthwise 3x3,
sigma=2, zero padding 1, normalised weights (rebuilt per call in the reference; here they are 9
kernel arguments).  The stencil is symmetric, so its backward is the same launch."""
import math

import torch
import torch.nn as nn

from .. import ops


def _gaussian_weights(kernel_size=3, sigma=2.0):
    """gaussian_blur.py:17-38 evaluated with the same fp32 torch expressions."""
    x_coord = torch.arange(kernel_size)
    x_grid = x_coord.repeat(kernel_size).view(kernel_size, kernel_size)
    y_grid = x_grid.t()
    xy_grid = torch.stack([x_grid, y_grid], dim=-1).float()
    mean = (kernel_size - 1) / 2.
    variance = sigma ** 2.
    k = (1. / (2. * math.pi * variance)) * torch.exp(-torch.sum((xy_grid - mean) ** 2., dim=-1) / (2 * variance))
    return k / torch.sum(k)


class _StencilFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w9):
        ctx.w9 = w9
        return ops.stencil3(x.float(), w9)

    @staticmethod
    def backward(ctx, g):
        return ops.stencil3(g.float(), ctx.w9[::-1]), None  # transpose of a correlation = flipped taps


class GaussianBlur(nn.Module):
    capturable = True    # deterministic launches: Hidden.enable_graph may capture a step through this layer

    def __init__(self, kernel_size=3, channels=3):
        super(GaussianBlur, self).__init__()
        if kernel_size != 3:
            raise NotImplementedError("the reference only instantiates the 3x3 blur")
        self.kernel_size = kernel_size
        self.channels = channels
        self.name = "G_Blur"
        self._w9 = _gaussian_weights(kernel_size).flatten().tolist()

    def get_gaussian_kernel(self, kernel_size=3, sigma=2, channels=3):
        return _gaussian_weights(self.kernel_size, sigma)

    def forward(self, tensor, cover_image=None):
        self.name = "GaussianBlur"
        if not tensor.is_cuda:
            raise RuntimeError("GaussianBlur runs on the HIP path only: move the input to cuda")
        return _StencilFn.apply(tensor, self._w9)

    def fwd(self, image):
        self.name = "GaussianBlur"
        return ops.stencil3(image, self._w9), None

    def bwd(self, ctx, g):
        return ops.stencil3(g, self._w9[::-1])
