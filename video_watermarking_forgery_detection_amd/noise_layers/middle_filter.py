"""MiddleBlur -- mirror of the reference's noise_layers/middle_filter.py:5-13, which wraps
kornia.filters.MedianBlur((k,k)).  kornia is a third-party dependency, unpinned by the reference and
absent from the build image: PARITY UNPINNED for this op.  Definition implemented (kornia's published
algorithm): per channel, zero padding k//2, median of the k*k window; the gradient goes to the
selected element.  k = 3 (IRNcrop_model.py:95) or 5 (IRNrhi_model.py:138)."""
import torch
import torch.nn as nn

from .. import ops


class _MedianFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        y, idx = ops.median_fwd(x.float(), k)
        ctx.k = k
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.median_bwd(g.float(), idx, ctx.k), None


class MiddleBlur(nn.Module):
    capturable = True    # deterministic launches: Hidden.enable_graph may capture a step through this layer

    def __init__(self, kernel):
        super(MiddleBlur, self).__init__()
        if kernel not in (3, 5):
            raise NotImplementedError("median kernel size 3 or 5")
        self.kernel = kernel
        self.name = "MiddleBlur"

    def forward(self, image):
        if not image.is_cuda:
            raise RuntimeError("MiddleBlur runs on the HIP path only: move the input to cuda")
        return _MedianFn.apply(image, self.kernel)

    def fwd(self, image):
        y, idx = ops.median_fwd(image, self.kernel)
        return y, idx

    def bwd(self, ctx, g):
        return ops.median_bwd(g, ctx, self.kernel)
