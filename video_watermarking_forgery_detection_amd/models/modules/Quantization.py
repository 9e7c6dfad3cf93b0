"""Quantization -- mirror of the reference's models/modules/Quantization.py:4-21:
round(255 x)/255 forward (no clamp: it is commented out at :8), identity gradient."""
import torch
import torch.nn as nn

from ... import ops


class Quant(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input):
        if not input.is_cuda:
            raise RuntimeError("Quantization runs on the HIP path only: move the input to cuda")
        return ops.quant(input)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class Quantization(nn.Module):
    def __init__(self):
        super(Quantization, self).__init__()
        self.name = "Quantization"

    def forward(self, input):
        return Quant.apply(input)

    def fwd(self, image):
        return ops.quant(image), None

    def bwd(self, ctx, g):
        return g
