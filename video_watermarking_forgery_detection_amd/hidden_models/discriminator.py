"""Discriminator -- mirror of the reference's hidden_models/discriminator.py:5-27
(`Discriminator(config)`, keys `before_linear.{i}.layers.{0,1}.*`, `linear.*`; returns logits,
the sigmoid is commented out in the reference at :26)."""
import torch
import torch.nn as nn

from .. import engine, ops
from ..options import HiDDenConfiguration
from .conv_bn_relu import ConvBNRelu
from .decoder import _StackLinearFn, bump_bn_counters, head_bwd, stack_bwd, stack_fwd, stack_fwd_loss


class Discriminator(nn.Module, engine.FlatModule):
    """Receives an image and decides whether it carries a watermark."""

    def __init__(self, config: HiDDenConfiguration):
        super(Discriminator, self).__init__()
        c = config.discriminator_channels
        self.channels = c
        layers = [ConvBNRelu(3, c)]
        for _ in range(config.discriminator_blocks - 1):
            layers.append(ConvBNRelu(c, c))
        layers.append(nn.AdaptiveAvgPool2d(output_size=(1, 1)))
        self.before_linear = nn.Sequential(*layers)
        self.linear = nn.Linear(c, 1)
        self.compute_dtype = torch.bfloat16

    def _blocks(self):
        return [m for m in self.before_linear if isinstance(m, ConvBNRelu)]

    def fwd(self, image, training=True):
        pooled, ctx = stack_fwd(self._blocks(), image, self.compute_dtype, training)
        out = ops.linear_head_fwd(pooled, self.linear.weight.data, self.linear.bias.data, self.channels)
        if training:
            engine.bump_bn_counters(self)
        return out, ctx

    def fwd_loss(self, image, label, gscale, grads, accumulate=False, gscale_dev=None):
        """fwd(image) + BCEWithLogitsLoss against the constant label (hidden.py:68-97) + the head's share of the backward (gscale * the
        loss's gradient): -> (logits [B,1], loss [1], ctx); continue with bwd(ctx, None, grads, accumulate, ...)"""
        out, loss, ctx = stack_fwd_loss(self, self._blocks(), image, self.compute_dtype, self.channels, 0, float(label), None, gscale,
                                        gscale_dev, grads, accumulate)
        engine.bump_bn_counters(self)
        return out, loss, ctx

    def bwd(self, ctx, g_out, grads, accumulate=False, need_input_grad=False, weight_grads=True, raw_input_grad=False):
        """weight_grads False (with need_input_grad): the gradient wrt the image only -- the convolutions' weight gradients are not
        computed (the generator's pass through the discriminator, hidden.py:85-103: nothing reads them)"""
        gvec = head_bwd(self, ctx, self.channels, g_out, grads, accumulate)
        return stack_bwd(self._blocks(), ctx, gvec, grads, accumulate, need_input_grad, weight_grads=weight_grads, raw_input_grad=raw_input_grad)

    def forward(self, image):
        if not image.is_cuda:
            raise RuntimeError("Discriminator runs on the HIP path only: move the module and input to cuda")
        return _StackLinearFn.apply(image, self, *self.parameters())
