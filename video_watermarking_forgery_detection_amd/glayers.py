"""General HIP layer toolkit for the networks either side of the HiDDeN path: nn.Conv2d / nn.ConvTranspose2d /
nn.Linear / the activations / residual and QF-attention combines / global average pooling / symmetric and replication padding /
spectral norm, each an autograd Function over the C ABI's wm_gconv_* / wm_unary_* / ... entry points (include/wm_hip.h, "general
layer family").  Parameters keep torch's layouts and names, so a reference state_dict loads unchanged
(models/networks.py:631-749, models/conditional_jpeg_generator.py:40-374,697-826).

Between layers an activation is a contiguous NHWC tensor [B,H,W,Cp] (Cp = channels rounded up to 16, padding channels zero or
ignored) of the network's compute dtype (float32 = the parity path, bfloat16 / float16); images enter and leave as NCHW float32
through `to_nhwc` / `to_nchw`.  There is no CPU or PyTorch fallback: every Function launches HIP kernels.
"""
import math

import weakref

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops

cpad = ops.cpad


def _pad_bias(bias, n):
    if bias is None:
        return None
    if bias.numel() == n and bias.dtype == torch.float32 and bias.is_contiguous():
        return bias.detach()            # (no padding channels: two launches fewer per call -- the embedder makes ~1,600 such calls a step)
    out = torch.zeros(n, device=bias.device, dtype=torch.float32)
    out[: bias.numel()] = bias.detach()
    return out


_PLAN = None
_FLAT_GRADS = {}      # parameter data_ptr -> (weakref flat parameters, weakref flat gradients, offset, numel) of a live FlatAdamW


def _flat_grad(ptr, numel):
    """the slice of a live FlatAdamW's gradient buffer that belongs to the parameter at address `ptr` (None if there is none: the
    optimiser was collected -- its addresses may have been reused -- or the tensor is not a parameter of one)"""
    e = _FLAT_GRADS.get(ptr)
    if e is None:
        return None
    flat, grad = e[0](), e[1]()
    if flat is None or grad is None or e[3] != numel or flat.data_ptr() + 4 * e[2] != ptr:
        if flat is None or grad is None:
            del _FLAT_GRADS[ptr]
        return None
    return grad[e[2]:e[2] + e[3]]


def set_pack_plan(plan):
    """Route the packed 3x3 weights of plain Conv2d layers through `plan` (ops.PackPlan; None switches it off): a layer's packed
    weight -- forward and transposed orientation -- is then a persistent tensor that ONE launch refreshes for every layer after an
    optimiser step (FlatAdamW.step does it; with another optimiser call plan.refresh() yourself) instead of one small launch per
    layer, call and orientation (2,800 a step in the invertible embedder).  In-place torch writes to a weight (load_state_dict, init)
    are noticed through its autograd version and fall back to a one-off pack until the next refresh."""
    global _PLAN
    _PLAN = plan


def _pack3(weight, rows, cols, dtype, transpose, plan_ok):
    w = weight.detach()
    if _PLAN is not None and plan_ok:
        return _PLAN.get(w, rows, cols, dtype, transpose=transpose)
    return ops.pack_w3x3(w, rows, cols, dtype, transpose=transpose)


def _fast3x3(weight, stride, pad, dtype, out_stride):
    """3x3 stride-1 pad-1 convolutions on 16-bit activations whose output stride is a multiple of 32 run on the hot path's
    wave-specialised / streamed-filter kernels (csrc/conv3x3*.hip, wgrad_ws.hip) instead of the general direct kernel"""
    return (dtype in (torch.bfloat16, torch.float16) and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and pad == 1 and out_stride % 32 == 0)


def _conv_forward(ctx, x, weight, bias, stride, pad):
    """the convolution itself + what both autograd nodes below keep for their backward (everything but the saved tensors)"""
    Cout, Cin, KH, KW = weight.shape
    B, IH, IW, KC = x.shape
    if KC != cpad(Cin):
        raise ValueError(f"conv expects {Cin} input channels (stride {cpad(Cin)}), the activation has stride {KC}")
    NC = cpad(Cout)
    OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
    ctx.geo = (stride, pad, bias is not None)
    ctx.plan_ok = isinstance(weight, nn.Parameter)       # (a computed weight -- spectral norm -- is a new tensor every call: never planned)
    ctx.bias_ptr = bias.data_ptr() if bias is not None else 0
    if _fast3x3(weight, stride, pad, x.dtype, NC):
        wp = _pack3(weight, NC, KC, x.dtype, False, ctx.plan_ok)
        out, _ = ops.conv3x3_fwd(x, wp, _pad_bias(bias, NC), None, None, want_stats=False)
        return out
    wp = ops.gconv_pack(weight.detach(), NC, KC, False, x.dtype)
    return ops.gconv_fwd(x, wp, _pad_bias(bias, NC), (OH, OW), KH, KW, stride, pad)


class _ConvFn(Function):
    """nn.Conv2d on NHWC: weight [Cout,Cin,KH,KW] f32"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        out = _conv_forward(ctx, x, weight, bias, stride, pad)
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.geo
        gx, gw, gb = _conv_backward(x, weight, g.contiguous(), stride, pad, ctx.needs_input_grad[0],
                                    ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]), has_bias, ctx.plan_ok, ctx.bias_ptr)
        return gx, gw, gb, None, None


def _conv_backward(x, weight, g, stride, pad, need_gx, need_gw, want_bias, plan_ok=False, bias_ptr=0):
    """(gx, gw, gb) of nn.Conv2d for the output gradient g (NHWC, contiguous); gb only if want_bias.  A parameter whose .grad is a view
    of a FlatAdamW buffer gets its gradient ADDED there by the kernel and None returned for it (autograd would otherwise launch one
    add per parameter and contribution: 3,200 a step in the invertible embedder, whose weights all serve twice)."""
    Cout, Cin, KH, KW = weight.shape
    gx = gw = gb = None
    if need_gx:
        if _fast3x3(weight, stride, pad, g.dtype, x.shape[3]):
            wt = _pack3(weight, g.shape[3], x.shape[3], g.dtype, True, plan_ok)
            gx, _ = ops.conv3x3_fwd(g, wt, None, None, None, want_stats=False)
        else:
            wt = ops.gconv_pack(weight.detach(), x.shape[3], g.shape[3], True, g.dtype)
            gx = ops.gconv_fwd(g, wt, None, (x.shape[1], x.shape[2]), KH, KW, stride, pad, dgrad=True)
    if need_gw:
        wacc = _flat_grad(weight.data_ptr(), weight.numel())
        bacc = _flat_grad(bias_ptr, Cout) if want_bias else None
        if wacc is None or (want_bias and bacc is None):
            wacc = bacc = None
        if _fast3x3(weight, stride, pad, g.dtype, 32):      # (the weight-gradient kernels take any 8-multiple of channels)
            gw = wacc.view(Cout, Cin, 3, 3) if wacc is not None else torch.empty(Cout, Cin, 3, 3, device=g.device, dtype=torch.float32)
            ops.conv3x3_wgrad(x, x.shape[3], None, None, g, gw, wacc is not None)
            gb = ops.gcolsum(g, Cout, out_acc=bacc) if want_bias else None
        else:
            gw, gb = ops.gconv_wgrad(g, x, Cout, Cin, KH, KW, stride, pad, want_bias=want_bias, dw_acc=wacc, db_acc=bacc)
        if wacc is not None:
            gw = gb = None
    return gx, gw, gb


FUSE_ELU = True   # (tests switch it off to compare with the separate launches)


def _elu_fused(kind, weight, stride, pad, x):
    return (FUSE_ELU and kind == "elu" and tuple(weight.shape[2:]) == (3, 3) and stride == 1 and pad == 1 and weight.shape[0] == 64
            and x.shape[3] == cpad(weight.shape[1]) and ops.conv3x3_fwd_elu_supported(x.shape[3], 64, x.dtype))


class _ConvEluFn(Function):
    """elu(nn.Conv2d(x)) for the 3x3 layers the persistent kernel takes (16-bit activations, 64 output channels, 16 / 32 / 64 input
    channels: the coupling subnets' conv1..conv4, models/invertible_net.py:326-366) in ONE launch forwards -- only the output is kept --
    and TWO + the slab reduction backwards: the input-gradient kernel forms gz = g * elu'(.) from (g, out) while it stages its tiles,
    writes gz for the weight gradient and leaves the bias gradient's partial sums, which the weight gradient's reduction launch folds
    (before: conv, ELU | ELU' + column sums, their reduce, input gradient, weight gradient, its reduce)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        Cout, Cin = weight.shape[0], weight.shape[1]
        ctx.plan_ok = isinstance(weight, nn.Parameter)
        ctx.bias_ptr = bias.data_ptr() if bias is not None else 0
        ctx.has_bias = bias is not None
        wp = _pack3(weight, 64, x.shape[3], x.dtype, False, ctx.plan_ok)
        out = ops.conv3x3_fwd_elu(x, wp, _pad_bias(bias, 64))
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        Cout, Cin = weight.shape[0], weight.shape[1]
        need_gx, need_gw, need_gb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        g = g.contiguous()
        if x.shape[3] in (64, 32, 16) and ops.conv3x3_dgrad_elufused_supported(x.shape[3], g.dtype):
            # (16-channel inputs: the filter packed to 32 rows -- the upper 16 zero -- for the 32-channel consumers, which store 16)
            wt = _pack3(weight, 64, max(x.shape[3], 32), g.dtype, True, ctx.plan_ok)
            gx, gz, part = ops.conv3x3_dgrad_elufused(g, out, wt, want_gz=need_gw, dx_stride=x.shape[3])
            gb = None
        else:   # (the 16-channel first layers: their input gradient runs on the general kernel)
            bacc = _flat_grad(ctx.bias_ptr, Cout) if need_gb else None
            gz, gb = ops.unary_bwd_colsum(out, g, "elu_out", Cout, db_acc=bacc)
            gx, gw, _ = _conv_backward(x, weight, gz, 1, 1, need_gx, need_gw, False, ctx.plan_ok)
            return (gx if need_gx else None), gw, (gb if need_gb and bacc is None else None)
        gw = None
        if need_gw or need_gb:
            wacc = _flat_grad(weight.data_ptr(), weight.numel())
            bacc = _flat_grad(ctx.bias_ptr, Cout) if need_gb else None
            if wacc is None or (need_gb and bacc is None):
                wacc = bacc = None
            gw = wacc.view(Cout, Cin, 3, 3) if wacc is not None else torch.empty(Cout, Cin, 3, 3, device=g.device, dtype=torch.float32)
            gb = bacc if bacc is not None else torch.empty(Cout, device=g.device, dtype=torch.float32)
            ops.conv3x3_wgrad_bias(x, gz, gw, wacc is not None, part, gb, bacc is not None)
            if wacc is not None:
                gw = gb = None
            elif not need_gb:
                gb = None
        return (gx if need_gx else None), gw, gb


class _ConvActFn(Function):
    """act(nn.Conv2d(x)) as ONE autograd node: the backward forms gz = g * act'(z) and the bias gradient (its column sums) in one pass
    over the data (ops.unary_bwd_colsum: two launches instead of three, gz not read back), then the convolution's two gradients from gz"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, kind):
        z = _conv_forward(ctx, x, weight, bias, stride, pad)
        ctx.save_for_backward(x, weight, z)
        ctx.kind = kind
        return ops.unary_fwd(z, kind)

    @staticmethod
    def backward(ctx, g):
        x, weight, z = ctx.saved_tensors
        stride, pad, has_bias = ctx.geo
        bacc = _flat_grad(ctx.bias_ptr, weight.shape[0]) if (has_bias and ctx.needs_input_grad[2]) else None
        gz, gb = ops.unary_bwd_colsum(z, g, ctx.kind, weight.shape[0], db_acc=bacc)
        gx, gw, _ = _conv_backward(x, weight, gz, stride, pad, ctx.needs_input_grad[0], ctx.needs_input_grad[1], False, ctx.plan_ok)
        return gx, gw, (gb if has_bias and ctx.needs_input_grad[2] and bacc is None else None), None, None, None


class _ConvTFn(Function):
    """nn.ConvTranspose2d on NHWC: weight [Cin,Cout,KH,KW] f32 -- the input gradient of the conv Cout -> Cin with the same filter"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        Cin, Cout, KH, KW = weight.shape
        B, IH, IW, KC = x.shape
        if KC != cpad(Cin):
            raise ValueError(f"transposed conv expects {Cin} input channels, the activation has stride {KC}")
        NC = cpad(Cout)
        OH, OW = (IH - 1) * stride - 2 * pad + KH, (IW - 1) * stride - 2 * pad + KW
        wt = ops.gconv_pack(weight.detach(), NC, KC, True, x.dtype)
        out = ops.gconv_fwd(x, wt, _pad_bias(bias, NC), (OH, OW), KH, KW, stride, pad, dgrad=True)
        ctx.save_for_backward(x, weight)
        ctx.geo = (stride, pad, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.geo
        Cin, Cout, KH, KW = weight.shape
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wp = ops.gconv_pack(weight.detach(), x.shape[3], g.shape[3], False, g.dtype)
            gx = ops.gconv_fwd(g, wp, None, (x.shape[1], x.shape[2]), KH, KW, stride, pad)
        if ctx.needs_input_grad[1]:
            gw, _ = ops.gconv_wgrad(x, g, Cin, Cout, KH, KW, stride, pad, want_bias=False)
        if has_bias and ctx.needs_input_grad[2]:
            gb = ops.gcolsum(g, Cout)
        return gx, gw, gb, None, None


class _ActFn(Function):
    @staticmethod
    def forward(ctx, x, kind):
        ctx.save_for_backward(x)
        ctx.kind = kind
        return ops.unary_fwd(x, kind)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.unary_bwd(x, g, ctx.kind), None


class _AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add_scaled(a, b, 1.0)

    @staticmethod
    def backward(ctx, g):
        return g, g


class _QFAttFn(Function):
    """x + gamma * res + beta, gamma / beta f32 [B, Cp] (conditional_jpeg_generator.py:196-200)"""

    @staticmethod
    def forward(ctx, x, res, gamma, beta):
        ctx.save_for_backward(res, gamma)
        return ops.qfatt_fwd(x, res, gamma, beta)

    @staticmethod
    def backward(ctx, g):
        res, gamma = ctx.saved_tensors
        gres, gg, gb = ops.qfatt_bwd(g, res, gamma)
        return g, gres, gg, gb


class _PoolFn(Function):
    """AdaptiveAvgPool2d((1,1)) + Flatten: NHWC [B,H,W,Cp] -> f32 [B,1,1,Cp]"""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (tuple(x.shape), x.dtype)
        return ops.gpool_fwd(x).view(x.shape[0], 1, 1, x.shape[3])

    @staticmethod
    def backward(ctx, g):
        shape, dtype = ctx.meta
        return ops.gpool_bwd(g.reshape(shape[0], shape[3]), shape, dtype)


class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, img, pads, mode, dtype):
        ctx.meta = (tuple(img.shape), pads, mode)
        return ops.pad_nchw_to_nhwc(img, pads, mode, dtype)

    @staticmethod
    def backward(ctx, g):
        shape, pads, mode = ctx.meta
        return ops.pad_nchw_to_nhwc_bwd(g, shape, pads, mode), None, None, None


class _ToNCHW(Function):
    @staticmethod
    def forward(ctx, x, C, H, W):
        ctx.meta = (tuple(x.shape), x.dtype)
        return ops.gunpack_nchw(x, C, H, W)

    @staticmethod
    def backward(ctx, g):
        shape, dtype = ctx.meta
        return ops.gunpack_nchw_bwd(g, shape, dtype), None, None, None


class _SpectralNormFn(Function):
    @staticmethod
    def forward(ctx, weight_orig, u, v, do_iter):
        wsn, sigma = ops.spectral_norm_fwd(weight_orig.detach(), u, v, do_iter)
        ctx.save_for_backward(wsn, u.clone(), v.clone(), sigma)
        return wsn

    @staticmethod
    def backward(ctx, g):
        wsn, u, v, sigma = ctx.saved_tensors
        return ops.spectral_norm_bwd(g, wsn, u, v, sigma), None, None, None


def to_nhwc(img, dtype, pads=(0, 0, 0, 0), mode=ops.PAD_SYMMETRIC):
    """NCHW f32 image -> NHWC activation, optionally padded (left, right, top, bottom) symmetrically or by replication"""
    if not img.is_cuda:
        raise RuntimeError("the HIP layer toolkit runs on the GPU only; there is no CPU fallback")
    return _ToNHWC.apply(img, tuple(int(p) for p in pads), mode, dtype)


def to_nchw(x, C, H=None, W=None):
    """NHWC activation -> NCHW f32 [B,C,H,W] (the top-left window when H / W are smaller than the activation's)"""
    return _ToNCHW.apply(x, C, x.shape[1] if H is None else H, x.shape[2] if W is None else W)


def add(a, b):
    return _AddFn.apply(a, b)


def qf_attention(x, res, gamma, beta):
    B, Cp = x.shape[0], x.shape[3]
    return _QFAttFn.apply(x, res, gamma.reshape(B, -1).float(), beta.reshape(B, -1).float())


def global_avg_pool(x):
    return _PoolFn.apply(x)


def _kaiming_uniform_(w, fan_in):
    # torch's default reset_parameters of Conv2d / Linear: kaiming_uniform_(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    with torch.no_grad():
        w.uniform_(-bound, bound)


class Conv2d(nn.Module):
    """nn.Conv2d(in, out, k, stride, padding, bias) on NHWC activations; parameters `weight` [out,in,k,k], `bias` [out]"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, self.kernel_size, self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        fan_in = in_channels * self.kernel_size ** 2
        _kaiming_uniform_(self.weight, fan_in)
        if bias:
            _kaiming_uniform_(self.bias, fan_in)

    def effective_weight(self):
        return self.weight

    def forward(self, x):
        return _ConvFn.apply(x, self.effective_weight(), self.bias, self.stride, self.padding)


class SpectralNormConv2d(Conv2d):
    """nn.utils.spectral_norm(nn.Conv2d(..)) (networks.py:1381-1385): parameters `weight_orig`, buffers `weight_u`, `weight_v`,
    one power iteration per training forward, none in eval."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias)
        w = self.weight
        del self._parameters["weight"]
        self.weight_orig = nn.Parameter(w.data)
        n = in_channels * self.kernel_size ** 2
        self.register_buffer("weight_u", nn.functional.normalize(torch.randn(out_channels), dim=0, eps=1e-12))
        self.register_buffer("weight_v", nn.functional.normalize(torch.randn(n), dim=0, eps=1e-12))

    def effective_weight(self):
        return _SpectralNormFn.apply(self.weight_orig, self.weight_u, self.weight_v, self.training)


class ConvAct(nn.Module):
    """nn.Sequential(conv, activation) with the state_dict keys of that Sequential (`0.weight`, `0.bias`) and one autograd node for the
    pair (_ConvActFn); `conv` is a Conv2d / SpectralNormConv2d of this module"""

    def __init__(self, conv, kind):
        super().__init__()
        if kind not in ops.ACT_KINDS:
            raise ValueError(f"unknown activation {kind}")
        self.add_module("0", conv)
        self.kind = kind

    def __getitem__(self, i):
        if i != 0:
            raise IndexError(i)
        return self._modules["0"]

    def forward(self, x):
        c = self._modules["0"]
        w = c.effective_weight()
        if _elu_fused(self.kind, w, c.stride, c.padding, x):
            return _ConvEluFn.apply(x, w, c.bias)
        return _ConvActFn.apply(x, w, c.bias, c.stride, c.padding, self.kind)

    def extra_repr(self):
        return self.kind


class FusedSequential(nn.Sequential):
    """nn.Sequential with the same children and state_dict keys whose forward runs every (Conv2d, Act) neighbour pair as one autograd
    node (_ConvActFn): the activation's backward and the convolution's bias gradient become one pass over the data"""

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], Act):
                w = m.effective_weight()
                if _elu_fused(mods[i + 1].kind, w, m.stride, m.padding, x):
                    x = _ConvEluFn.apply(x, w, m.bias)
                else:
                    x = _ConvActFn.apply(x, w, m.bias, m.stride, m.padding, mods[i + 1].kind)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d(in, out, k, stride, padding, bias); `weight` [in,out,k,k]"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = int(kernel_size), int(stride), int(padding)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, self.kernel_size, self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        fan_in = out_channels * self.kernel_size ** 2      # torch computes fan_in from dim 1 of the weight
        _kaiming_uniform_(self.weight, fan_in)
        if bias:
            _kaiming_uniform_(self.bias, fan_in)

    def forward(self, x):
        return _ConvTFn.apply(x, self.weight, self.bias, self.stride, self.padding)


class Linear(nn.Module):
    """nn.Linear on [B,1,1,Cp] activations (a 1x1 convolution); `weight` [out,in], `bias` [out]"""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        _kaiming_uniform_(self.weight, in_features)
        if bias:
            _kaiming_uniform_(self.bias, in_features)

    def forward(self, x):
        return _ConvFn.apply(x, self.weight.view(self.out_features, self.in_features, 1, 1), self.bias, 1, 0)


class Act(nn.Module):
    """ReLU / LeakyReLU(0.2) / GELU / ELU / Sigmoid / Tanh"""

    def __init__(self, kind):
        super().__init__()
        if kind not in ops.ACT_KINDS:
            raise ValueError(f"unknown activation {kind}")
        self.kind = kind

    def forward(self, x):
        return _ActFn.apply(x, self.kind)

    def extra_repr(self):
        return self.kind


class GlobalAvgPool(nn.Module):
    """torch.nn.AdaptiveAvgPool2d((1,1)); the result is f32 [B,1,1,Cp] and stays f32 through the Linear layers after it"""

    def forward(self, x):
        return global_avg_pool(x)


class Flatten(nn.Module):
    """torch.nn.Flatten after the pool: nothing to do on [B,1,1,Cp]; keeps the reference's Sequential indices"""

    def forward(self, x):
        return x


def vector_in(v, dtype=torch.float32):
    """[B,F] f32 -> [B,1,1,cpad(F)] activation (the input of a Linear stack)"""
    if not v.is_cuda:
        raise RuntimeError("the HIP layer toolkit runs on the GPU only; there is no CPU fallback")
    B, F = v.shape
    return to_nhwc(v.reshape(B, F, 1, 1).float(), dtype)


def vector_out(x, F):
    """[B,1,1,Cp] -> [B,F] f32"""
    return to_nchw(x, F, 1, 1).reshape(x.shape[0], F)


# ----------------------------------------------------------------------------- invertible-embedder pieces (models/invertible_net.py)
class _HaarFn(Function):
    @staticmethod
    def forward(ctx, x, C, fac, up, by_wavelet=False):
        ctx.meta = (C, fac, up, by_wavelet)
        return ops.haar(x, C, fac, up, by_wavelet)

    @staticmethod
    def backward(ctx, g):
        C, fac, up, by_wavelet = ctx.meta
        return ops.haar(g.contiguous(), C, fac, not up, by_wavelet), None, None, None, None


class _ChanSliceFn(Function):
    """x[..., off:off+n] as its own NHWC tensor (stride cpad(n), padding zero); one launch each way (ops.chan_place)"""

    @staticmethod
    def forward(ctx, x, off, n):
        ctx.meta = (x.shape[3], off, n)
        return ops.chan_place(cpad(n), x, off, 0, n)

    @staticmethod
    def backward(ctx, g):
        stride, off, n = ctx.meta
        return ops.chan_place(stride, g.contiguous(), 0, off, n), None, None


class _ChanCatFn(Function):
    """torch.cat((a[..., :na], b[..., :nb]), channel dim)"""

    @staticmethod
    def forward(ctx, a, na, b, nb):
        ctx.meta = (na, nb)
        return ops.chan_place(cpad(na + nb), a, 0, 0, na, b, 0, na, nb)

    @staticmethod
    def backward(ctx, g):
        na, nb = ctx.meta
        g = g.contiguous()
        return ops.chan_place(cpad(na), g, 0, 0, na), None, ops.chan_place(cpad(nb), g, na, 0, nb), None


class _CouplingFn(Function):
    """rev False: e(s) * x + t; rev True: (x - t) / e(s)   (invertible_net.py:140-141,153-173)"""

    @staticmethod
    def forward(ctx, x, s, t, clamp, eps, rev):
        y = ops.coupling_fwd(x, s, t, clamp, eps, rev)
        ctx.save_for_backward(y if rev else x, s)
        ctx.meta = (clamp, eps, rev)
        return y

    @staticmethod
    def backward(ctx, g):
        v, s = ctx.saved_tensors
        clamp, eps, rev = ctx.meta
        gx, gs, gt = ops.coupling_bwd(g, v, s, clamp, eps, rev)
        return gx, gs, gt, None, None, None


def haar_down(x, C, fac, by_wavelet=False):
    return _HaarFn.apply(x, C, float(fac), False, bool(by_wavelet))


def haar_up(x, C, fac, by_wavelet=False):
    return _HaarFn.apply(x, C, float(fac), True, bool(by_wavelet))


def chan_slice(x, off, n):
    return _ChanSliceFn.apply(x, off, n)


def chan_cat(a, na, b, nb):
    return _ChanCatFn.apply(a, na, b, nb)


def coupling(x, s, t, clamp, eps, rev):
    return _CouplingFn.apply(x, s, t, float(clamp), float(eps), bool(rev))


# ----------------------------------------------------------------------------- losses, clamp and the optimiser of the literal IRNrhi step
def _scaled(grad, g):
    """grad * g for the 1-element upstream gradient g of a scalar loss (device scalar: no host sync)"""
    return ops.scale_dev_(grad.clone(), g.reshape(1).float().contiguous())


class _SmoothL1Fn(Function):
    @staticmethod
    def forward(ctx, a, b, beta):
        loss, grad = ops.smooth_l1(a, b, beta, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.shape = a.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, g).reshape(ctx.shape), None, None


class _BCEProbFn(Function):
    @staticmethod
    def forward(ctx, p, target):
        loss, grad = ops.bce_prob(p, target, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.shape = p.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, g).reshape(ctx.shape), None


class _CrossEntropyFn(Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, grad = ops.cross_entropy(logits, labels, want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, g), None


class _Clamp01Fn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.clamp01_fwd(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.clamp01_bwd(x, g)


def smooth_l1_loss(a, b, beta=1.0):
    """nn.SmoothL1Loss()(a, b); b carries no gradient (a target)"""
    return _SmoothL1Fn.apply(a, b.detach(), float(beta))


def bce_loss(p, target):
    """nn.BCELoss()(p, torch.full_like(p, target)) for the constant targets 1.0 / 0.0 of the GAN terms"""
    return _BCEProbFn.apply(p, float(target))


def cross_entropy_loss(logits, labels):
    return _CrossEntropyFn.apply(logits, labels)


def clamp01(x):
    """torch.clamp(x, 0, 1) (gradient where 0 <= x <= 1)"""
    return _Clamp01Fn.apply(x)


class FlatAdamW:
    """torch.optim.AdamW over ONE flat f32 buffer holding every trainable parameter of `module` (each nn.Parameter becomes a view of
    it, each .grad a view of a flat gradient buffer autograd accumulates into): zero_grad = one memset, clip_grad_norm_ = one norm,
    step = one launch of the library's Adam kernel with decoupled weight decay.  While the optimiser lives, the convolution backward
    kernels ADD a registered parameter's gradient straight into the flat buffer and hand autograd None for it (_conv_backward): use
    loss.backward(); torch.autograd.grad(loss, parameters) would see None for those parameters and still change the buffer."""

    def __init__(self, module, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        ps = [p for p in module.parameters() if p.requires_grad]
        if not ps or not all(p.is_cuda and p.dtype == torch.float32 for p in ps):
            raise RuntimeError("FlatAdamW: parameters must be float32 CUDA tensors (move the module to the GPU first)")
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.params, o = ps, 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + k].view(p.shape)
                p.grad = self.grad[o:o + k].view(p.shape)
                _FLAT_GRADS[p.data_ptr()] = (weakref.ref(self.flat), weakref.ref(self.grad), o, k)
                o += k
        self.lr, self.betas, self.eps, self.weight_decay, self.t = lr, betas, eps, weight_decay, 0

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:           # autograd may have replaced a view (first accumulation into a None grad)
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or p.grad.data_ptr() >= self.grad.data_ptr() + 4 * self.grad.numel():
                raise RuntimeError("FlatAdamW: a parameter's .grad no longer views the flat gradient buffer (do not set grads to None)")

    def clip_grad_norm_(self, max_norm):
        """nn.utils.clip_grad_norm_(module.parameters(), max_norm); returns the [2] device tensor (coefficient, total norm)"""
        return ops.clip_grad_norm_([self.grad], float(max_norm))

    def step(self):
        self.t += 1
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t, decoupled=True)
        if _PLAN is not None:
            _PLAN.refresh()        # every registered packed weight from the new values, one launch


class CapturedStep:
    """A launch-bound step (the invertible embedder issues ~20,000 kernels of 3-20 us per forward + reverse + backward) captured ONCE into a
    hipGraph and replayed: `fn()` must be shape-static, read its inputs from tensors that are updated in place (copy_), and not synchronise
    with the host (no .item() / float()).  Gradients land in the parameters' existing .grad tensors (FlatAdamW's flat buffer), so
    zero_grad belongs INSIDE fn and the optimiser step -- whose bias correction depends on the host's step count -- outside:

        step = CapturedStep(lambda: fwd_bwd(x_static))      # two eager warm-up calls on a side stream, then the capture
        for batch in data: x_static.copy_(batch); step.replay(); opt.step()

    step.result holds what fn returned at capture (tensors that every replay overwrites).  Hand results out through that return value, or
    .detach() what fn stores elsewhere: an fn that keeps the autograd graph of its previous call alive (it rebinds an outer name, a dict
    entry, an attribute ... to a tensor that requires grad) is REFUSED with a RuntimeError before anything is captured.  Why: a graph that
    outlives its call keeps the AccumulateGrad node of every parameter it used alive, and the next call's graph picks those nodes up
    again, so they never die -- and a node runs on the stream that was current when it was CREATED.  When the first call ran eagerly on
    the default stream, the captured backward therefore makes the legacy default stream wait on an event of the capturing stream;
    CUDA invalidates such a capture with an error, hipStreamEndCapture on ROCm 7.2 ended it in a segmentation fault
    (gpurun_out/inn3.log, round 2; tools/dbg_capture.py).  The check: before the last warm-up call the AccumulateGrad node of every CUDA
    parameter is tagged (Node.metadata); a node with no graph holding it dies with the tag, so a tag that is still there after the call's
    result has been dropped proves a leaked graph."""

    def __init__(self, fn, warmup=2):
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedStep: GPU only")
        import gc
        warmup = max(int(warmup), 2)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        tag, key = object(), "wm_captured_step_probe"
        params = []

        def accumulator(p):
            return torch.autograd.graph.get_gradient_edge(p).node

        with torch.cuda.stream(side):
            for i in range(warmup):
                if i == warmup - 1:
                    gc.collect()
                    params = [o for o in gc.get_objects() if isinstance(o, torch.nn.Parameter) and o.is_cuda and o.requires_grad]
                    for p in params:
                        accumulator(p).metadata[key] = tag
                r = fn()
                del r
            gc.collect()
            leaked = [p for p in params if accumulator(p).metadata.get(key) is tag]
        torch.cuda.current_stream().wait_stream(side)
        if leaked:
            raise RuntimeError(
                f"CapturedStep: fn keeps the autograd graph of its previous call alive (the AccumulateGrad nodes of {len(leaked)} of {len(params)} "
                "parameters survive the call): it stores a tensor that requires grad outside itself (an outer variable, a dict entry, an "
                "attribute).  Return such tensors from fn (step.result) or .detach() them; capturing this fn can crash the process inside "
                "hipStreamEndCapture")
        self.graph = torch.cuda.CUDAGraph()
        # (the cyclic collector stays off while the stream captures: a dead cycle it frees may own device memory or another hipGraph, and
        # releasing those is illegal during a capture -- hidden_models/hidden.py::_StepGraph)
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.result = fn()
        finally:
            if gc_was_on:
                gc.enable()

    def replay(self):
        self.graph.replay()
        return self.result
