"""Crop -- mirror of the reference's noise_layers/crop.py:8-55 (`forward`): random (numpy RNG) or
`apex`-given rectangle, bilinear resize back to the full frame; returns (image, apex).  The slice and
the interpolation are one gather kernel (the rectangle is an offset into the source planes)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


class _CropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rect):
        x = x.float()
        ctx.rect, ctx.hw = rect, (x.shape[2], x.shape[3])
        return ops.resample_fwd(x, rect, (x.shape[2], x.shape[3]), ops.BILINEAR)

    @staticmethod
    def backward(ctx, g):
        return ops.resample_bwd(g.float(), None, ctx.hw, ctx.rect, ops.BILINEAR), None


class Crop(nn.Module):
    def __init__(self):
        super(Crop, self).__init__()
        self.name = "Crop"

    def get_random_rectangle_inside(self, image_shape, height_ratio, width_ratio):
        """crop.py:13-30"""
        image_height, image_width = image_shape[2], image_shape[3]
        remaining_height = int(height_ratio * image_height)
        remaining_width = int(width_ratio * image_width)
        height_start = 0 if remaining_height == image_height else np.random.randint(0, image_height - remaining_height)
        width_start = 0 if remaining_width == image_width else np.random.randint(0, image_width - remaining_width)
        return height_start, height_start + remaining_height, width_start, width_start + remaining_width

    def _apex(self, image, apex, min_rate, max_rate):
        # the ratio draws happen even when apex is given (crop.py:33-40): the RNG stream must advance the same way
        if min_rate:
            self.height_ratio = min_rate + (max_rate - min_rate) * np.random.rand()
            self.width_ratio = min_rate + (max_rate - min_rate) * np.random.rand()
        else:
            self.height_ratio = 0.3 + 0.7 * np.random.rand()
            self.width_ratio = 0.3 + 0.7 * np.random.rand()
        self.height_ratio = min(self.height_ratio, self.width_ratio + 0.2)
        self.width_ratio = min(self.width_ratio, self.height_ratio + 0.2)
        if apex is not None:
            return tuple(int(v) for v in apex)
        return self.get_random_rectangle_inside(image.shape, self.height_ratio, self.width_ratio)

    def forward(self, image, apex=None, min_rate=0.5, max_rate=1.0):
        if not image.is_cuda:
            raise RuntimeError("Crop runs on the HIP path only: move the input to cuda")
        h_start, h_end, w_start, w_end = self._apex(image, apex, min_rate, max_rate)
        rect = (h_start, h_end - h_start, w_start, w_end - w_start)
        return _CropFn.apply(image, rect), (h_start, h_end, w_start, w_end)

    def fwd(self, image, apex=None, min_rate=0.5, max_rate=1.0):
        h_start, h_end, w_start, w_end = self._apex(image, apex, min_rate, max_rate)
        rect = (h_start, h_end - h_start, w_start, w_end - w_start)
        y = ops.resample_fwd(image, rect, (image.shape[2], image.shape[3]), ops.BILINEAR)
        self.last_apex = (h_start, h_end, w_start, w_end)
        return y, (rect, (image.shape[2], image.shape[3]))

    def bwd(self, ctx, g):
        rect, hw = ctx
        return ops.resample_bwd(g, None, hw, rect, ops.BILINEAR)
