"""Encoder -- mirror of the reference's hidden_models/encoder.py:7-43 on the HIP kernels.

Same ctor (`Encoder(config)`), attributes and state_dict keys (`conv_layers.{i}.layers.{0,1}.*`,
`after_concat_layer.layers.*`, `final_layer.*`).  The concat of encoder.py:40 is
cat([message, features, image]); in HBM it is laid out as [features(64) | message(L) | image(3) | 0-pad]
so the feature slice stays 16-byte aligned, and the after-concat weights are permuted to match when
they are packed -- the arithmetic is the reference's.
"""
import torch
import torch.nn as nn

from .. import engine, ops
from ..options import HiDDenConfiguration
from .conv_bn_relu import ConvBNRelu


class EncCtx:
    __slots__ = ("layers", "cat_ctx", "final_in", "B", "H", "W", "split", "message", "image")


class Encoder(nn.Module, engine.FlatModule):
    """Inserts a watermark into an image."""

    def __init__(self, config: HiDDenConfiguration):
        super(Encoder, self).__init__()
        self.H = config.H
        self.W = config.W
        self.conv_channels = config.encoder_channels
        self.num_blocks = config.encoder_blocks
        self.message_length = config.message_length
        layers = [ConvBNRelu(3, self.conv_channels)]
        for _ in range(config.encoder_blocks - 1):
            layers.append(ConvBNRelu(self.conv_channels, self.conv_channels))
        self.conv_layers = nn.Sequential(*layers)
        self.after_concat_layer = ConvBNRelu(self.conv_channels + 3 + config.message_length, self.conv_channels)
        self.final_layer = nn.Conv2d(self.conv_channels, 3, kernel_size=1)
        self.compute_dtype = torch.bfloat16
        c, L = self.conv_channels, config.message_length
        assert c % 8 == 0
        # packed position of reference concat channel ci (reference order: message, features, image)
        self._cat_ld = engine.round_up(c + L + 3, 16)
        self._perm = [c + i for i in range(L)] + list(range(c)) + [c + L + i for i in range(3)]
        self._perm_dev = None
        # the same layer WITHOUT the concat (16-bit dtypes, 64 feature channels): conv64(features) + [conv3(image) + message bias];
        # reference channel -> position in the 64-channel feature operand / the 16-channel image operand (>= the width: not there)
        DROP = 1 << 20
        self._fperm = [DROP] * L + list(range(c)) + [DROP] * 3
        self._iperm = [DROP] * (L + c) + [0, 1, 2]
        self._fperm_dev = self._iperm_dev = None
        self.split_concat = True   # A/B switch: False builds the 97(112)-channel tensor like round 1

    # -- explicit engine path ------------------------------------------------------------------
    def fwd(self, image, message, training=True):
        """image [B,3,H,W] f32 cuda, message [B,L] -> (encoded [B,3,H,W] f32, ctx)"""
        if image.shape[2] != self.H or image.shape[3] != self.W:
            raise RuntimeError(f"Encoder configured for {self.H}x{self.W}, got {tuple(image.shape[2:])}")  # encoder.py:37
        dt = self.compute_dtype
        B = image.shape[0]
        c, L = self.conv_channels, self.message_length
        ctx = EncCtx()
        ctx.layers = []
        a = engine.image_to_act(image, dt)
        for blk in self.conv_layers:
            a, cx = engine.cbr_forward(blk.layers[0], blk.layers[1], a, dt, training=training)
            ctx.layers.append(cx)
        blk = self.after_concat_layer
        ctx.split = (self.split_concat and dt in (torch.bfloat16, torch.float16) and c == 64 and a.t.shape[-1] == 64 and a.scale is not None
                     and self.H >= 2 and self.W >= 2)
        if ctx.split:
            # encoder.py:34-41 without materialising the concat: P = conv3(image) + bias + message term (one store-bound pass),
            # added to conv64(features) in the epilogue of the ordinary 64 -> 64 kernel, before the BatchNorm statistics
            conv, bn = blk.layers[0], blk.layers[1]
            P = ops.concat_side_fwd(image, conv.weight.data, conv.bias.data, message, dt, 0, L, L + c)
            wp = engine._packed(conv, 64, 64, dt, self._fperm, False)
            d = engine._opposite(a.rev)
            y, st = ops.conv3x3_fwd_addin(a.t, wp, a.scale, a.shift, P, reverse=d)
            a5, ctx.cat_ctx = engine.cbr_finish(conv, bn, a, y, st, d, None, training)
            ctx.message, ctx.image = message, image
        else:
            cat = torch.empty(B, self.H, self.W, self._cat_ld, device=image.device, dtype=dt)
            ops.concat_full(a.t, a.scale, a.shift, message, image, cat, c)   # [features | message | image | 0-pad], one pass
            a5, ctx.cat_ctx = engine.cbr_forward(blk.layers[0], blk.layers[1], engine.Act(cat, c + L + 3), dt,
                                                 perm=self._perm, training=training)
        ctx.final_in = a5
        fl = self.final_layer
        if training and engine.sharing_image_acts():
            # inside a training step the encoded image is read by the discriminator (twice) and, under an Identity attack, the decoder:
            # the head writes their 16-channel NHWC input beside the NCHW result
            enc, a16 = ops.conv1x1_head_fwd(a5.t, a5.scale, a5.shift, fl.weight.data.view(3, c), fl.bias.data, act=0, want_act16=True)
            engine.register_image_act(enc, a16)
        else:
            enc = ops.conv1x1_head_fwd(a5.t, a5.scale, a5.shift, fl.weight.data.view(3, c), fl.bias.data, act=0)
        if training:
            engine.bump_bn_counters(self)
        return enc, ctx

    def body_param_count(self):
        """number of parameters of conv_layers = offset of after_concat_layer in the flat buffers (parameters() order)"""
        return sum(p.numel() for p in self.conv_layers.parameters())

    def bwd(self, ctx, g_enc, grads, accumulate=False, after_head=None):
        """g_enc [B,3,H,W] f32: gradient wrt the encoded image.  Parameter gradients go to `grads`.
        (No gradient wrt image/message: the reference's inputs do not require grad.)
        after_head: optional callable run once the gradients of final_layer and after_concat_layer are queued (data parallel:
        their bucket leaves while the body layers run their backward)."""
        c = self.conv_channels
        if self._perm_dev is None or self._perm_dev.device != g_enc.device:
            self._perm_dev = torch.tensor(self._perm, dtype=torch.int32, device=g_enc.device)
        fl = self.final_layer
        a5 = ctx.final_in
        # the head's backward also reduces the after-concat layer's BatchNorm-backward sums (y and g are in its registers anyway)
        g, bnp = ops.conv1x1_head_bwd(a5.t, a5.scale, a5.shift, fl.weight.data.view(3, c), g_enc,
                                      grads[fl.weight].view(3, c), grads[fl.bias], accumulate, want_bn_partials=True)
        if bnp is not None:
            a5.bwd = (g, bnp, None, g._version)
        blk = self.after_concat_layer
        if ctx.split:
            g = self._after_concat_bwd_split(ctx, g, grads, accumulate)
        else:
            g = engine.cbr_backward(blk.layers[0], blk.layers[1], ctx.cat_ctx, grads, g=g, accumulate=accumulate,
                                    dgrad_channels=c, perm_dev=self._perm_dev)
        if after_head is not None:
            after_head()
        n = len(self.conv_layers)
        for i in range(n - 1, -1, -1):
            blk = self.conv_layers[i]
            g = engine.cbr_backward(blk.layers[0], blk.layers[1], ctx.layers[i], grads, g=g, accumulate=accumulate,
                                    need_input_grad=(i > 0))
        return None

    def _after_concat_bwd_split(self, ctx, g, grads, accumulate):
        """backward of the after-concat ConvBNRelu in its split form.  The feature part is an ordinary 64 -> 64 layer (fused input
        gradient: BatchNorm-backward apply + the sums of the last body layer; weight gradient with that layer's finalisation riding
        on its slab reduction) writing the feature channels of dW; the image channels of dW come from the image-fed weight-gradient
        kernel, the message channels from per-sample border-class sums of dy."""
        conv, bn = self.after_concat_layer.layers[0], self.after_concat_layer.layers[1]
        cx = ctx.cat_ctx
        y, x, dt = cx.y, cx.x, cx.y.dtype
        dev = y.device
        if self._fperm_dev is None or self._fperm_dev.device != dev:
            self._fperm_dev = torch.tensor(self._fperm, dtype=torch.int32, device=dev)
            self._iperm_dev = torch.tensor(self._iperm, dtype=torch.int32, device=dev)
        dw = grads[conv.weight]
        pre, cx.out.bwd = cx.out.bwd, None
        if pre is not None and pre[0] is g and g._version == pre[3]:    # sums already reduced by the head's backward
            coef = ops.bn_bwd_coef_raw(pre[1], y, cx.stats, 64, bn.weight.data, grads[bn.weight], grads[bn.bias], accumulate)
        else:
            coef = ops.bn_bwd_coef(g, None, y, cx.stats, 64, bn.weight.data, grads[bn.weight], grads[bn.bias], accumulate)
        wpt = engine._packed(conv, 64, 64, dt, self._fperm, True)
        d = engine._opposite(getattr(g, "_wm_rev", None))
        fused = ops.conv3x3_dgrad_applyfused_supported(64, 64, dt)
        feed = fused and ops.conv3x3_dgrad_bwdstats_supported(64, 64, dt) and x.src is not None
        if feed:
            dy, gx, part = ops.conv3x3_dgrad_applyfused(g, y, cx.stats, coef, wpt, x.t, x.scale, x.shift, reverse=d)
            pcoef = ops.conv3x3_wgrad(x.t, 64, x.scale, x.shift, dy, dw, accumulate, perm_dev=self._fperm_dev, reverse=not d,
                                      fin=engine.fin_rider(x, part, grads, accumulate))
            x.bwd = (gx, part, pcoef, gx._version)
        else:
            if fused:
                dy, gx, _ = ops.conv3x3_dgrad_applyfused(g, y, cx.stats, coef, wpt, reverse=d)
                ops.conv3x3_wgrad(x.t, 64, x.scale, x.shift, dy, dw, accumulate, perm_dev=self._fperm_dev, reverse=not d)
            else:   # (debug build with the fusion switched off: the stand-alone apply pass, as engine.cbr_backward's generic branch)
                dy = ops.bn_bwd(g, None, y, cx.stats, 64, bn.weight.data, grads[bn.weight], grads[bn.bias], accumulate, None, coef=coef)
                ops.conv3x3_wgrad(x.t, 64, x.scale, x.shift, dy, dw, accumulate, perm_dev=self._fperm_dev, reverse=True)
                if x.src is not None and ops.conv3x3_dgrad_bwdstats_supported(64, 64, dt):
                    gx, part = ops.conv3x3_dgrad_bwdstats(dy, wpt, x.t, x.scale, x.shift)
                    x.bwd = (gx, part, None, gx._version)
                else:
                    gx, _ = ops.conv3x3_fwd(dy, wpt, None, None, None, want_stats=False)
                d = False
        gx._wm_rev = d
        img16 = engine.image_to_act(ctx.image, dt)
        ops.conv3x3_wgrad(img16.t, 16, None, None, dy, dw, accumulate, perm_dev=self._iperm_dev)
        ops.concat_side_msg_wgrad(dy, ctx.message, dw, accumulate, 0, self.message_length)
        return gx

    def _bump_bn_counters(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None:
                m.num_batches_tracked += 1

    # -- reference-style call (autograd) -------------------------------------------------------
    def forward(self, image, message):
        if not image.is_cuda:
            raise RuntimeError("Encoder runs on the HIP path only: move the module and inputs to cuda")
        return _EncoderFn.apply(image, message, self, *self.parameters())


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, message, mod, *params):
        enc, c = mod.fwd(image.float().contiguous(), message.float().reshape(message.shape[0], -1), training=mod.training)
        ctx.c, ctx.mod = c, mod
        return enc

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        n = sum(p.numel() for p in mod.parameters())
        flat = torch.zeros(n, device=g.device, dtype=torch.float32)
        grads = engine.grad_dict(mod, flat)
        mod.bwd(ctx.c, g.float().contiguous(), grads, accumulate=False)
        return (None, None, None) + tuple(grads[p] for p in mod.parameters())
