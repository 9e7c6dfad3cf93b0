"""Resize -- mirror of the reference's noise_layers/resize.py:15-55: bicubic down/up-sample to
int(r*H) x int(r*W) and back, clamp to [0,1]; r ~ U(0.5,1.5) from np.random.rand unless
`resize_ratio` is given."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


def random_float(min, max):
    return np.random.rand() * (max - min) + min


_KIND = {"bicubic": ops.BICUBIC, "bilinear": ops.BILINEAR}


class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nh, nw, kind):
        x = x.float()
        H, W = x.shape[2], x.shape[3]
        small = ops.resample_fwd(x, (0, H, 0, W), (nh, nw), kind)
        y = ops.resample_fwd(small, (0, nh, 0, nw), (H, W), kind, clamp01=True)
        ctx.dims = (H, W, nh, nw, kind)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        H, W, nh, nw, kind = ctx.dims
        gs = ops.resample_bwd(g.float(), y, (nh, nw), (0, nh, 0, nw), kind)
        return ops.resample_bwd(gs, None, (H, W), (0, H, 0, W), kind), None, None, None


class Resize(nn.Module):
    def __init__(self, resize_ratio_range=(0.5, 1.5), interpolation_method='bicubic'):
        super(Resize, self).__init__()
        self.name = "Resize"
        self.resize_ratio_min = resize_ratio_range[0]
        self.resize_ratio_max = resize_ratio_range[1]
        self.interpolation_method = interpolation_method
        if interpolation_method not in _KIND:
            raise NotImplementedError("interpolation_method must be bicubic or bilinear")

    def _sizes(self, x, resize_ratio):
        # the reference names shape[2] "width" (resize.py:31); only the order matters
        if resize_ratio is None:
            resize_ratio = random_float(self.resize_ratio_min, self.resize_ratio_max)
        return int(resize_ratio * x.shape[2]), int(resize_ratio * x.shape[3])

    def forward(self, noised_image, resize_ratio=None):
        self.name = "Resize"
        if not noised_image.is_cuda:
            raise RuntimeError("Resize runs on the HIP path only: move the input to cuda")
        nh, nw = self._sizes(noised_image, resize_ratio)
        return _ResizeFn.apply(noised_image, nh, nw, _KIND[self.interpolation_method])

    def fwd(self, image, resize_ratio=None):
        kind = _KIND[self.interpolation_method]
        H, W = image.shape[2], image.shape[3]
        nh, nw = self._sizes(image, resize_ratio)
        small = ops.resample_fwd(image, (0, H, 0, W), (nh, nw), kind)
        y = ops.resample_fwd(small, (0, nh, 0, nw), (H, W), kind, clamp01=True)
        return y, (y, H, W, nh, nw, kind)

    def bwd(self, ctx, g):
        y, H, W, nh, nw, kind = ctx
        gs = ops.resample_bwd(g, y, (nh, nw), (0, nh, 0, nw), kind)
        return ops.resample_bwd(gs, None, (H, W), (0, H, 0, W), kind)
