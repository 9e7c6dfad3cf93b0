"""UNet -- the tamper-localisation head; mirror of the reference's network/UNet.py:7-97 on the HIP
kernels (same ctor `UNet(in_channels=3, out_channels=1, init_features=32)`, same state_dict keys:
`encoder1.enc1conv1.weight`, `encoder1.enc1norm1.*`, `upconv4.*`, `decoder4.dec4conv1.*`, `conv.*`).

Data flow in HBM (NHWC, bf16 or f32):
  * every 3x3 conv output stays RAW; its BatchNorm+ReLU is applied by the consumer kernel on load;
  * MaxPool2d reads the raw map once and writes the pooled activations AND the activated
    full-resolution skip straight into the right half of the decoder's concat buffer;
  * ConvTranspose2d writes its pixel-shuffled output into the left half of that buffer
    (torch.cat((dec, enc), dim=1) of UNet.py:54 is never a separate pass);
  * the 1x1 conv + sigmoid head emits the mask as NCHW f32 like the reference.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import engine, ops


class _UCtx:
    pass


class UNet(nn.Module, engine.FlatModule):
    def __init__(self, in_channels=3, out_channels=1, init_features=32):
        super(UNet, self).__init__()
        if in_channels != 3 or out_channels not in (1, 3):
            raise NotImplementedError("the HIP UNet path covers the reference's use: UNet(3, 1, features)")
        features = init_features
        if features % 32 != 0:
            raise NotImplementedError("init_features must be a multiple of 32")
        self.features = features
        self.encoder1 = UNet._block(in_channels, features, name="enc1")
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder2 = UNet._block(features, features * 2, name="enc2")
        self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder3 = UNet._block(features * 2, features * 4, name="enc3")
        self.pool3 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder4 = UNet._block(features * 4, features * 8, name="enc4")
        self.pool4 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.bottleneck = UNet._block(features * 8, features * 16, name="bottleneck")
        self.upconv4 = nn.ConvTranspose2d(features * 16, features * 8, kernel_size=2, stride=2)
        self.decoder4 = UNet._block((features * 8) * 2, features * 8, name="dec4")
        self.upconv3 = nn.ConvTranspose2d(features * 8, features * 4, kernel_size=2, stride=2)
        self.decoder3 = UNet._block((features * 4) * 2, features * 4, name="dec3")
        self.upconv2 = nn.ConvTranspose2d(features * 4, features * 2, kernel_size=2, stride=2)
        self.decoder2 = UNet._block((features * 2) * 2, features * 2, name="dec2")
        self.upconv1 = nn.ConvTranspose2d(features * 2, features, kernel_size=2, stride=2)
        self.decoder1 = UNet._block(features * 2, features, name="dec1")
        self.conv = nn.Conv2d(in_channels=features, out_channels=out_channels, kernel_size=1)
        self.compute_dtype = torch.bfloat16

    @staticmethod
    def _block(in_channels, features, name):
        # parameter containers only (identical keys / init to the reference); the compute is in fwd/bwd
        return nn.Sequential(OrderedDict([
            (name + "conv1", nn.Conv2d(in_channels, features, kernel_size=3, padding=1, bias=False)),
            (name + "norm1", nn.BatchNorm2d(num_features=features)),
            (name + "relu1", nn.ReLU(inplace=True)),
            (name + "conv2", nn.Conv2d(features, features, kernel_size=3, padding=1, bias=False)),
            (name + "norm2", nn.BatchNorm2d(num_features=features)),
            (name + "relu2", nn.ReLU(inplace=True)),
        ]))

    @staticmethod
    def _parts(block):
        m = list(block.children())
        return (m[0], m[1]), (m[3], m[4])

    def _block_fwd(self, block, a, training):
        (c1, n1), (c2, n2) = self._parts(block)
        dt = self.compute_dtype
        a1, x1 = engine.cbr_forward(c1, n1, a, dt, training=training)
        a2, x2 = engine.cbr_forward(c2, n2, a1, dt, training=training)
        return a2, (x1, x2)

    def _block_bwd(self, block, ctxs, g, grads, accumulate, need_input_grad=True):
        (c1, n1), (c2, n2) = self._parts(block)
        g = engine.cbr_backward(c2, n2, ctxs[1], grads, g=g, accumulate=accumulate)
        return engine.cbr_backward(c1, n1, ctxs[0], grads, g=g, accumulate=accumulate, need_input_grad=need_input_grad)

    # -- explicit engine path --------------------------------------------------------------------
    def fwd(self, x, training=True):
        """x [B,3,H,W] f32 cuda (H, W multiples of 16) -> (sigmoid mask [B,out,H,W] f32, ctx)"""
        B, _, H, W = x.shape
        if H % 16 or W % 16:
            raise RuntimeError(f"UNet needs H, W divisible by 16 (4 poolings), got {H}x{W}")
        dt = self.compute_dtype
        f = self.features
        ctx = _UCtx()
        ctx.enc, ctx.cat, ctx.dec, ctx.encact = [], [], [], []
        a = engine.image_to_act(x, dt)
        h, w = H, W
        for lvl in (1, 2, 3, 4):
            C = f * (1 << (lvl - 1))
            a, cx = self._block_fwd(getattr(self, f"encoder{lvl}"), a, training)
            cat = torch.empty(B, h, w, 2 * C, device=x.device, dtype=dt)   # [upconv half | skip half]
            pooled = ops.bnrelu_maxpool2(a.t, a.scale, a.shift, C, act_out=cat, act_c0=C)
            ctx.enc.append(cx)
            ctx.encact.append(a)
            ctx.cat.append(cat)
            a = engine.Act(pooled, C)
            h, w = h // 2, w // 2
        a, ctx.bott = self._block_fwd(self.bottleneck, a, training)
        for lvl in (4, 3, 2, 1):
            C = f * (1 << (lvl - 1))
            up = getattr(self, f"upconv{lvl}")
            cat = ctx.cat[lvl - 1]
            ops.upconv2x2_fwd(a.t, a.scale, a.shift, up.weight.data, up.bias.data, cat, 0)
            ctx.dec.append((a, None))
            a, cx = self._block_fwd(getattr(self, f"decoder{lvl}"), engine.Act(cat, 2 * C), training)
            ctx.dec[-1] = (ctx.dec[-1][0], cx)
        ctx.last = a
        oc = self.conv.weight.shape[0]
        out = ops.conv1x1_head_fwd(a.t, a.scale, a.shift, self.conv.weight.data.view(oc, f), self.conv.bias.data, act=1)
        ctx.out = out
        if training:
            engine.bump_bn_counters(self)
        return out, ctx

    def grad_buckets(self):
        """four contiguous [lo, hi) ranges of the flat gradient buffer in the order the backward completes them (reverse
        layer order; flat order = parameters() order: encoder1-4, bottleneck, upconv4, decoder4, ..., decoder1, conv):
        [upconv3 .. conv], [upconv4, decoder4], [bottleneck], [encoder1-4] -- 0.75 / 2.3 / 3.5 / 1.2 M parameters (3 / 9 / 14 / 5 MB) at features=32"""
        off, start = 0, {}
        for name, mod in self.named_children():
            start[name] = off
            off += sum(p.numel() for p in mod.parameters())
        return [(start["upconv3"], off), (start["upconv4"], start["upconv3"]), (start["bottleneck"], start["upconv4"]),
                (0, start["bottleneck"])]

    def bwd(self, ctx, g_out, grads, accumulate=False, need_input_grad=True, g_is_logit=False, bucket_ready=None):
        """g_out: gradient wrt the sigmoid output [B,out,H,W] f32 (g_is_logit: wrt the 1x1 conv's output, the sigmoid already
        chained by the loss kernel) -> gradient wrt x [B,3,H,W] (or None).
        bucket_ready: optional callable(lo, hi) run when the gradients of flat range [lo, hi) (grad_buckets()) are queued."""
        bk = self.grad_buckets() if bucket_ready is not None else None
        f = self.features
        oc = self.conv.weight.shape[0]
        y = ctx.out
        g_logit = g_out if g_is_logit else g_out * y * (1.0 - y)  # sigmoid' on the small mask tensor
        a = ctx.last
        g, bnp = ops.conv1x1_head_bwd(a.t, a.scale, a.shift, self.conv.weight.data.view(oc, f), g_logit,
                                      grads[self.conv.weight].view(oc, f), grads[self.conv.bias], accumulate, want_bn_partials=True)
        if bnp is not None:   # the last decoder layer's BatchNorm-backward sums come out of the head's backward: no reduce pass
            a.bwd = (g, bnp, None, g._version)
        gcats = {}
        for k, lvl in enumerate((1, 2, 3, 4)):  # decoder blocks were run 4,3,2,1 -> undo 1,2,3,4
            C = f * (1 << (lvl - 1))
            a_in, cx = ctx.dec[3 - k]
            gcat = self._block_bwd(getattr(self, f"decoder{lvl}"), cx, g, grads, accumulate)   # [B,h,w,2C]
            gcats[lvl] = gcat
            up = getattr(self, f"upconv{lvl}")
            g = ops.upconv2x2_bwd(a_in.t, a_in.scale, a_in.shift, up.weight.data, gcat, 0, grads[up.weight], grads[up.bias], accumulate)
            if bk is not None and lvl in (3, 4):
                bucket_ready(*bk[lvl - 3])
        g = self._block_bwd(self.bottleneck, ctx.bott, g, grads, accumulate)
        if bk is not None:
            bucket_ready(*bk[2])
        for lvl in (4, 3, 2, 1):
            C = f * (1 << (lvl - 1))
            a = ctx.encact[lvl - 1]
            g = ops.maxpool2_bwd(a.t, a.scale, a.shift, g, gcats[lvl], C, C)
            g = self._block_bwd(getattr(self, f"encoder{lvl}"), ctx.enc[lvl - 1], g, grads, accumulate,
                                need_input_grad=(lvl > 1 or need_input_grad))
        if bk is not None:
            bucket_ready(*bk[3])
        if not need_input_grad:
            return None
        return ops.nhwc_to_nchw(g, 3, 0)

    # -- reference-style call (autograd) ---------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("UNet runs on the HIP path only: move the module and input to cuda")
        return _UNetFn.apply(x, self, *self.parameters())


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, *params):
        out, c = mod.fwd(x.float().contiguous(), training=mod.training)
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
