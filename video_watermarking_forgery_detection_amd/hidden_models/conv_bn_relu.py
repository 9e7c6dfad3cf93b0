"""ConvBNRelu -- mirror of the reference's hidden_models/conv_bn_relu.py:3-18
(same class name, ctor arguments and state_dict keys `layers.0.*`, `layers.1.*`), computed by the
HIP kernels: implicit-GEMM conv3x3 on MFMA with the BatchNorm statistics in its epilogue.

`forward(x)` keeps the reference contract (NCHW f32 in, NCHW f32 out, autograd works); inside the
training step the stacks are driven through engine.cbr_forward/cbr_backward instead, which never
materialises the activated tensor.
"""
import torch
import torch.nn as nn

from .. import engine, ops


class _CBRFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, weight, bias, gamma, beta):
        dtype = mod.compute_dtype
        B, C, H, W = x.shape
        CinX = engine.round_up(C, 16)
        xt = torch.empty(B, H, W, CinX, device=x.device, dtype=dtype)
        ops.nchw_to_nhwc(x.float(), xt, 0, CinX - C)
        conv, bn = mod.layers[0], mod.layers[1]
        act, c = engine.cbr_forward(conv, bn, engine.Act(xt, C), dtype, training=mod.training)
        if mod.training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        ctx.c, ctx.mod, ctx.C = c, mod, C
        Cout = weight.shape[0]
        out = torch.empty(B, H, W, act.t.shape[-1], device=x.device, dtype=dtype)
        ops.bnrelu_copy(act.t, act.scale, act.shift, out, 0, act.t.shape[-1])
        return ops.nhwc_to_nchw(out, Cout, 0)

    @staticmethod
    def backward(ctx, gout):
        mod, c = ctx.mod, ctx.c
        conv, bn = mod.layers[0], mod.layers[1]
        dtype = mod.compute_dtype
        B, Cout, H, W = gout.shape
        CoutP = c.y.shape[-1]
        g = torch.empty(B, H, W, CoutP, device=gout.device, dtype=dtype)
        ops.nchw_to_nhwc(gout.float(), g, 0, CoutP - Cout)
        tmp = {p: torch.empty_like(p.data) for p in (conv.weight, bn.weight, bn.bias)}
        tmp[conv.bias] = torch.zeros_like(conv.bias.data)   # identically zero in front of a training-mode BatchNorm
        gx = engine.cbr_backward(conv, bn, c, tmp, g=g, need_input_grad=ctx.needs_input_grad[0])
        gin = ops.nhwc_to_nchw(gx, ctx.C, 0) if gx is not None else None
        return gin, None, tmp[conv.weight], tmp[conv.bias], tmp[bn.weight], tmp[bn.bias]


class ConvBNRelu(nn.Module):
    """Conv3x3(stride 1, pad 1, bias) + BatchNorm2d + ReLU on the MI355X kernels."""

    def __init__(self, channels_in, channels_out, stride=1):
        super(ConvBNRelu, self).__init__()
        if stride != 1:
            raise NotImplementedError("the HIP conv path implements stride 1 only (all the reference uses)")
        # plain torch modules as parameter containers: identical keys / init to the reference
        self.layers = nn.Sequential(
            nn.Conv2d(channels_in, channels_out, 3, stride, padding=1),
            nn.BatchNorm2d(channels_out),
            nn.ReLU(inplace=True),
        )
        self.compute_dtype = torch.bfloat16

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("ConvBNRelu runs on the HIP path only: move the module and input to cuda")
        conv, bn = self.layers[0], self.layers[1]
        return _CBRFunction.apply(x, self, conv.weight, conv.bias, bn.weight, bn.bias)
