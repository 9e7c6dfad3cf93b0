"""Decoder and Discriminator share one shape: a ConvBNRelu stack on an image, a global average
pool, a Linear.  This file mirrors the reference's hidden_models/decoder.py:6-35
(`Decoder(config)`, keys `layers.{i}.layers.{0,1}.*`, `linear.*`)."""
import torch
import torch.nn as nn

from .. import engine, ops
from ..options import HiDDenConfiguration
from .conv_bn_relu import ConvBNRelu


class StackCtx:
    __slots__ = ("layers", "pooled", "pool_stats", "HW", "x_is_leaf_image", "gvec", "head_coef", "g_out")


def stack_fwd(blocks, image, dt, training):
    """image [B,3,H,W] f32 -> (pooled [B,CP] f32 of the last block's ReLU output, ctx)"""
    ctx = StackCtx()
    ctx.layers = []
    a = engine.image_to_act(image, dt)
    for blk in blocks:
        a, cx = engine.cbr_forward(blk.layers[0], blk.layers[1], a, dt, training=training)
        ctx.layers.append(cx)
    if training and ops.pool_stats_enabled():
        ctx.pooled, ctx.pool_stats = ops.bnrelu_avgpool_stats(a.t, a.scale, a.shift)
    else:
        ctx.pooled, ctx.pool_stats = ops.bnrelu_avgpool(a.t, a.scale, a.shift), None
    ctx.HW = image.shape[2] * image.shape[3]
    ctx.gvec = ctx.head_coef = ctx.g_out = None
    return ctx.pooled, ctx


def stack_fwd_loss(mod, blocks, image, dt, I, kind, target, messages, gscale, gscale_dev, grads, accumulate):
    """training forward of the stack + its pooled head + the LOSS on the head's output + everything of the backward that depends on nothing
    else -- the head's weight gradients, the per-sample gradient vector of the pooled features, the pooled layer's BatchNorm-backward
    coefficients -- in one launch behind the pool (ops.pooled_head; hidden.py:68-101 calls criterion(D(x), label) / mse(decoder(x), m)
    and .backward() right after).  kind 0: BCEWithLogits against the constant `target`; kind 1: the message loss against `messages`.
    -> (head output [B,O], loss tensor [1] or [2], ctx); the backward continues with mod.bwd(ctx, None, ...).  Falls back to the
    separate launches (same results) when the sizes exceed the one-workgroup form or the pool kept no statistics."""
    pooled, ctx = stack_fwd(blocks, image, dt, True)
    lin = mod.linear
    O = lin.weight.shape[0]
    last = ctx.layers[-1]
    B, CP = pooled.shape
    conv, bn = blocks[-1].layers[0], blocks[-1].layers[1]
    if ctx.pool_stats is not None and last.stats.is_contiguous() and ops.pooled_head_supported(B, CP, I, O) and pooled._base is not None:
        out, loss, ctx.gvec, ctx.head_coef = ops.pooled_head(
            pooled._base, I, lin.weight.data, lin.bias.data, kind, target, messages, gscale, gscale_dev, grads[lin.weight], grads[lin.bias],
            accumulate, 1.0 / ctx.HW, conv.weight.shape[0], B * ctx.HW, bn.weight.data, last.stats, grads[bn.weight], grads[bn.bias])
        return out, loss, ctx
    out = ops.linear_head_fwd(pooled, lin.weight.data, lin.bias.data, I)
    if kind == 0:
        loss, g = ops.bce_logits(out, target, gscale, gscale_dev=gscale_dev)
    else:
        loss, g = ops.message_loss(out, messages, gscale, gscale_dev=gscale_dev)
    ctx.g_out = g.view_as(out)
    return out, loss, ctx


def head_bwd(mod, ctx, I, g_out, grads, accumulate):
    """nn.Linear after the pool, backward: fills the weight / bias gradients and returns the per-sample gradient vector
    [B,CP] of the last ConvBNRelu's pooled output (already / (H*W)), one launch.  (A ctx of stack_fwd_loss has it all already.)"""
    if ctx.gvec is not None:
        if g_out is not None:
            raise RuntimeError("this forward already ran its head's backward (fwd_loss): call bwd(ctx, None, ...)")
        return ctx.gvec
    if g_out is None:
        g_out = ctx.g_out
    CP = ctx.layers[-1].y.shape[-1]
    return ops.linear_head_bwd(ctx.pooled, mod.linear.weight.data, g_out, grads[mod.linear.weight], grads[mod.linear.bias],
                               accumulate, CP, 1.0 / ctx.HW)


def stack_bwd(blocks, ctx, gvec, grads, accumulate, need_input_grad, weight_grads=True, raw_input_grad=False):
    """gvec [B,CP] f32 (gradient wrt the pooled features / (H*W), zero padded) -> gradient wrt the image [B,3,H,W] or None
    (weight_grads False: engine.cbr_backward's input-gradient-only form; raw_input_grad: the first layer's NHWC input gradient as its
    kernel wrote it -- channels 0..2 are the image's -- for a consumer that converts it on the way, ops.image_grad_mse)"""
    g = None
    n = len(blocks)
    for i in range(n - 1, -1, -1):
        blk = blocks[i]
        g = engine.cbr_backward(blk.layers[0], blk.layers[1], ctx.layers[i], grads, g=g, gvec=gvec if i == n - 1 else None,
                                accumulate=accumulate, need_input_grad=(i > 0 or need_input_grad),
                                pool_stats=ctx.pool_stats if i == n - 1 else None, weight_grads=weight_grads,
                                coef_pre=ctx.head_coef if i == n - 1 else None)
    if not need_input_grad:
        return None
    return g if raw_input_grad else ops.nhwc_to_nchw(g, 3, 0)


def bump_bn_counters(mod):
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None:
            m.num_batches_tracked += 1


class _StackLinearFn(torch.autograd.Function):
    """reference-style autograd entry for Decoder / Discriminator"""

    @staticmethod
    def forward(ctx, image, mod, *params):
        out, c = mod.fwd(image.float().contiguous(), training=mod.training)
        ctx.c, ctx.mod = c, mod
        return out

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        n = sum(p.numel() for p in mod.parameters())
        flat = torch.zeros(n, device=g.device, dtype=torch.float32)
        grads = engine.grad_dict(mod, flat)
        gin = mod.bwd(ctx.c, g.float().contiguous(), grads, accumulate=False, need_input_grad=ctx.needs_input_grad[0])
        return (gin, None) + tuple(grads[p] for p in mod.parameters())


class Decoder(nn.Module, engine.FlatModule):
    """Receives a (noised) watermarked image and extracts the message."""

    def __init__(self, config: HiDDenConfiguration):
        super(Decoder, self).__init__()
        self.channels = config.decoder_channels
        self.message_length = config.message_length
        layers = [ConvBNRelu(3, self.channels)]
        for _ in range(config.decoder_blocks - 1):
            layers.append(ConvBNRelu(self.channels, self.channels))
        layers.append(ConvBNRelu(self.channels, config.message_length))
        layers.append(nn.AdaptiveAvgPool2d(output_size=(1, 1)))
        self.layers = nn.Sequential(*layers)
        self.linear = nn.Linear(config.message_length, config.message_length)
        self.compute_dtype = torch.bfloat16

    def _blocks(self):
        return [m for m in self.layers if isinstance(m, ConvBNRelu)]

    def fwd(self, image, training=True):
        pooled, ctx = stack_fwd(self._blocks(), image, self.compute_dtype, training)
        L = self.message_length
        out = ops.linear_head_fwd(pooled, self.linear.weight.data, self.linear.bias.data, L)
        if training:
            engine.bump_bn_counters(self)
        return out, ctx

    def fwd_loss(self, image, messages, gscale, grads, accumulate=False, gscale_dev=None):
        """fwd(image) + the message loss (hidden.py:96-99,109-111) + the head's share of the backward: -> (decoded [B,L], [2] = MSE, bitwise
        error, ctx); continue with bwd(ctx, None, grads, accumulate)"""
        out, loss, ctx = stack_fwd_loss(self, self._blocks(), image, self.compute_dtype, self.message_length, 1, 0.0, messages, gscale,
                                        gscale_dev, grads, accumulate)
        engine.bump_bn_counters(self)
        return out, loss, ctx

    def bwd(self, ctx, g_out, grads, accumulate=False, need_input_grad=True):
        gvec = head_bwd(self, ctx, self.message_length, g_out, grads, accumulate)
        return stack_bwd(self._blocks(), ctx, gvec, grads, accumulate, need_input_grad)

    def forward(self, image_with_wm):
        if not image_with_wm.is_cuda:
            raise RuntimeError("Decoder runs on the HIP path only: move the module and input to cuda")
        return _StackLinearFn.apply(image_with_wm, self, *self.parameters())
