"""Explicit forward / backward of the ConvBNRelu stacks on the HIP kernels.

This is the host side of the hot path: it sequences the C-ABI kernels (ops.py)
for HiDDeN's encoder / decoder / discriminator (hidden_models/*.py of the
reference) without autograd -- the backward order is written out, so the step
can be captured, overlapped with RCCL and profiled kernel by kernel.

Data layout in HBM
  * activations: NHWC, bf16 (production) or f32 (parity), one tensor per conv
    output holding the RAW convolution result y; the BatchNorm+ReLU that follows
    is never materialised -- an `Act` carries (y, scale, shift) and every consumer
    kernel applies relu(scale*y+shift) while it stages its input;
  * image inputs: 3 channels zero-padded to 16 (one 32-byte bf16 pixel);
  * parameters / gradients / Adam moments: f32, one flat buffer per network.
"""
import contextlib
import threading

import torch

from . import ops


class Act:
    """NHWC activation: logical value = t (if scale is None) or relu(scale*t+shift)."""
    __slots__ = ("t", "C", "scale", "shift", "bwd", "rev", "src")

    def __init__(self, t, C, scale=None, shift=None, rev=None, src=None):
        self.t, self.C, self.scale, self.shift = t, C, scale, shift
        self.bwd = None   # backward: (g, partials, coef or None, g._version) left by the consumer's dgrad when it reduced this layer's BatchNorm sums
        self.src = src    # (bn module, stats [4,CP]) of the ConvBNRelu that produced t (lets the consumer's backward finish this layer's statistics).
                          # NOT the layer's ctx: ctx.out -> Act -> src -> ctx would be a reference cycle, and a cycle keeps a step's 134 MB
                          # tensors alive until Python's cyclic collector runs -- the caching allocator then falls back to hipMalloc mid-step
                          # (measured: 40-60 ms host stalls)
        self.rev = rev    # sweep direction of the conv that just wrote t (False forward, True backward, None: not fresh)


def _opposite(rev):
    """sweep direction for a kernel whose main input was written in direction `rev`: start where the producer stopped (its
    last tiles are the part of the tensor still in the Infinity Cache).  None (not fresh) -> forward."""
    return rev is False


def round_up(n, m):
    return (n + m - 1) // m * m


_tls = threading.local()   # per host thread: .image_acts = id(image) -> (image, version, dtype, NHWC16 tensor) while a share_image_acts()
                            # block is open; .bumps = module -> pending count while a defer_bn_counters() block is open


@contextlib.contextmanager
def share_image_acts(dtype=None):
    """inside the block, image_to_act converts an image tensor once: a training step feeds `images` to the discriminator and the
    encoder and `encoded` to two discriminator passes (the images are not modified in between -- checked through their
    version counters).  dtype: the networks' activation dtype, for producers that can write the NHWC16 form themselves
    (wanted_image_act_dtype)"""
    if getattr(_tls, "image_acts", None) is not None:
        yield
        return
    _tls.image_acts = {}
    _tls.image_act_dtype = dtype
    try:
        yield
    finally:
        _tls.image_acts = None
        _tls.image_act_dtype = None


def image_to_act(img, dtype):
    """[B,3,H,W] f32 (NCHW) -> Act over a [B,H,W,16] zero-padded NHWC tensor."""
    B, C, H, W = img.shape
    assert C == 3
    cache = getattr(_tls, "image_acts", None)
    if cache is not None:
        hit = cache.get(id(img))
        if hit is not None and hit[0] is img and hit[1] == img._version and hit[2] == dtype:
            return Act(hit[3], 3)
    t = torch.empty(B, H, W, 16, device=img.device, dtype=dtype)
    ops.nchw_to_nhwc(img, t, 0, 13)
    if cache is not None:
        cache[id(img)] = (img, img._version, dtype, t)
    return Act(t, 3)


def register_image_act(img, t):
    """inside a share_image_acts() block: `t` [B,H,W,16] IS image_to_act(img, t.dtype) already (written by the kernel that produced img:
    the encoder's head, the block-JPEG attack), so the first layer that reads img launches no conversion"""
    cache = getattr(_tls, "image_acts", None)
    if cache is not None and t is not None:
        cache[id(img)] = (img, img._version, t.dtype, t)


def sharing_image_acts():
    return getattr(_tls, "image_acts", None) is not None


def wanted_image_act_dtype():
    """inside a training step (share_image_acts(dtype)): the activation dtype of the networks that will read an image this step produces --
    a kernel that writes such an image may write its NHWC16 form beside it (register_image_act); None outside a step"""
    return getattr(_tls, "image_act_dtype", None) if sharing_image_acts() else None


class CBRCtx:
    __slots__ = ("x", "y", "stats", "perm", "training", "out")


def _packed(conv, CoutP, CinP, dtype, perm, transpose):
    """packed weights of `conv`: from the owning network's PackPlan (one launch per optimiser step, see
    FlatModule.refresh_packs) when there is one, else packed on the spot."""
    plan = getattr(conv, "_wm_plan", None)
    if plan is None:
        return ops.pack_w3x3(conv.weight.data, CoutP, CinP, dtype, perm=perm, transpose=transpose)
    # (the Parameter itself, not a `.data` alias: its version counter is the one torch bumps for `with no_grad(): w.add_(..)` etc.)
    return plan.get(conv.weight.detach(), CoutP, CinP, dtype, perm=perm, transpose=transpose)


def cbr_forward(conv, bn, x, dtype, perm=None, training=True, momentum=0.1):
    """ConvBNRelu forward (conv_bn_relu.py:11-15).  conv.weight [Cout,Cin,3,3], conv.bias or None,
    bn: BatchNorm2d parameters/buffers.  x: Act whose physical channel count (x.t.shape[-1]) is the
    K extent.  Returns (Act(y, scale, shift), ctx)."""
    Cout = conv.weight.shape[0]
    CoutP = round_up(Cout, 32)
    CinX = x.t.shape[-1]
    wp = _packed(conv, CoutP, CinX, dtype, perm, False)
    bias = conv.bias.data if conv.bias is not None else None
    d = _opposite(x.rev)
    y, st = ops.conv3x3_fwd(x.t, wp, bias, x.scale, x.shift, want_stats=training, reverse=d)
    return cbr_finish(conv, bn, x, y, st, d, perm, training, momentum)


def cbr_finish(conv, bn, x, y, st, d, perm=None, training=True, momentum=0.1):
    """second half of a ConvBNRelu forward: the raw conv output y (+ its statistics partials st) -> (Act(y, scale, shift), ctx)"""
    Cout = conv.weight.shape[0]
    CoutP = y.shape[-1]
    B, H, W, _ = y.shape
    if training:
        stats = ops.bn_finalize(st, Cout, CoutP, B * H * W, bn.weight.data, bn.bias.data, bn.running_mean,
                                bn.running_var, momentum if bn.momentum is None else bn.momentum, bn.eps)
    else:
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = bn.weight.data * invstd
        stats = torch.zeros(4, CoutP, device=y.device, dtype=torch.float32)
        stats[0, :Cout] = scale
        stats[1, :Cout] = bn.bias.data - bn.running_mean * scale
        stats[2, :Cout] = bn.running_mean
        stats[3, :Cout] = invstd
    ctx = CBRCtx()
    ctx.x, ctx.y, ctx.stats, ctx.perm, ctx.training = x, y, stats, perm, training
    ctx.out = Act(y, Cout, stats[0], stats[1], rev=d, src=(bn, stats))
    return ctx.out, ctx


def fin_rider(x, part, grads, accumulate):
    """the BatchNorm-backward finalisation of the layer that produced Act `x`, to ride on a weight-gradient slab reduction"""
    if x.src is None or not ops.fin_rider_enabled():
        return None
    pbn, pstats = x.src
    return dict(partials=part, y_shape=tuple(x.t.shape), stats=pstats, C=x.C, gamma=pbn.weight.data, dgamma=grads[pbn.weight],
                dbeta=grads[pbn.bias], accumulate=accumulate)


def cbr_backward(conv, bn, ctx, grads, g=None, gvec=None, need_input_grad=True, accumulate=False,
                 dgrad_channels=None, perm_dev=None, pool_stats=None, weight_grads=True, coef_pre=None):
    """Backward of ConvBNRelu.  g: NHWC gradient wrt the ReLU output ([B,H,W,>=CoutP]) or gvec [B,CoutP]
    (global-average-pool gradient, already / (H*W)).  grads: dict param -> f32 grad view.
    Returns the NHWC gradient wrt the (activated) input, `dgrad_channels` wide (default: the input's
    physical channels rounded up to 32), or None.
    weight_grads=False: the input gradient alone -- conv.weight's gradient is neither computed nor touched (the BatchNorm affine
    gradients, a by-product of the backward coefficients, still are): the discriminator under the generator's loss, whose parameter
    gradients nobody reads (hidden.py:67 zeroes them first thing in the next step)."""
    if not ctx.training:
        raise RuntimeError("backward through eval-mode BatchNorm is not supported by the HIP path")
    y, x = ctx.y, ctx.x
    Cout = conv.weight.shape[0]
    dtype = y.dtype
    # The gradient of a conv bias in front of a training-mode BatchNorm is identically zero (sum(dy) == 0: the batch mean
    # removes the bias); the reference's autograd produces rounding noise of ~1e-9 there.  The bias gradient view is
    # zero-initialised and never written, which is the exact value and saves a reduction + two launches per layer.
    dbias = None
    # The dgrad that produced g (the next layer's, below) may have reduced this layer's sums in its epilogue already
    pre, ctx.out.bwd = ctx.out.bwd, None
    d = _opposite(getattr(g, "_wm_rev", None)) if g is not None else False   # this layer's input-gradient sweep

    def tag(gx, direction=d):
        gx._wm_rev = direction
        return gx
    gam, dgam, dbet = bn.weight.data, grads[bn.weight], grads[bn.bias]

    def coef_of():
        if gvec is not None and coef_pre is not None:     # finished already by the fused pooled head (ops.pooled_head: dgamma / dbeta written there)
            return coef_pre
        if gvec is not None and pool_stats is not None:   # (N+, S+) from the forward pool: no pass over y
            return ops.bn_bwd_coef_pooled(gvec, pool_stats, y, ctx.stats, Cout, gam, dgam, dbet, accumulate)
        if pre is not None:
            # the consumer's input-gradient kernel already reduced this layer's sums over the tensor it returned: they are only
            # valid for exactly that tensor, unmodified (a different or edited g would silently use stale sums, or -- when the
            # rider already wrote dgamma / dbeta -- count them twice)
            if g is None or pre[0] is not g or g._version != pre[3]:
                raise RuntimeError("cbr_backward: the gradient handed to this layer is not the tensor its consumer's input-gradient "
                                   "kernel produced (or it was modified in place since); its pre-reduced BatchNorm sums are stale")
            if pre[2] is not None:   # finished already, as a rider on the next layer's weight-gradient reduction (dgamma / dbeta written there)
                return pre[2]
            return ops.bn_bwd_coef_raw(pre[1], y, ctx.stats, Cout, gam, dgam, dbet, accumulate)
        return ops.bn_bwd_coef(g, gvec, y, ctx.stats, Cout, gam, dgam, dbet, accumulate)

    # An image-fed first layer whose input needs no gradient: dy has ONE consumer, the weight gradient -- its kernel forms dy
    # from (g, y) while staging the tile, so the apply pass (a write and a read of dy) disappears.
    if (weight_grads and not need_input_grad and g is not None and x.scale is None and perm_dev is None and ctx.stats.is_contiguous()
            and g.shape[-1] == y.shape[-1] and ops.conv3x3_wgrad_bnfused_supported(x.t.shape[-1], y.shape[-1], dtype)):
        ops.conv3x3_wgrad_bnfused(x.t, g, y, ctx.stats, coef_of(), grads[conv.weight], accumulate)
        return None
    rows = dgrad_channels or round_up(x.t.shape[-1], 32)
    # the input is itself a ConvBNRelu output: this layer's dgrad can reduce THAT layer's BatchNorm-backward sums on the way
    feed_stats = (need_input_grad and x.scale is not None and perm_dev is None and ctx.perm is None and rows == 64
                  and x.t.shape[-1] == 64 and x.t.is_contiguous() and ops.conv3x3_dgrad_bwdstats_supported(y.shape[-1], rows, dtype))
    def rider(part):
        """the feeding layer's BatchNorm-backward finalisation, to ride on this layer's weight-gradient slab reduction"""
        if x.src is None or not ops.fin_rider_enabled():
            return None
        pbn, pstats = x.src
        return dict(partials=part, y_shape=tuple(x.t.shape), stats=pstats, C=x.C, gamma=pbn.weight.data, dgamma=grads[pbn.weight],
                    dbeta=grads[pbn.bias], accumulate=accumulate)

    if not weight_grads:
        if not need_input_grad:
            coef_of()      # dgamma / dbeta only
            return None
        rows16 = ctx.perm is None and perm_dev is None and ctx.stats.is_contiguous() and y.shape[-1] == 64 and rows in (64, 32)
        if gvec is not None and rows16 and rows == 64 and feed_stats and ops.conv3x3_gvfused_supported(64, 64, dtype):
            gx, part = ops.conv3x3_dgrad_bwdstats(y, _packed(conv, 64, 64, dtype, None, True), x.t, x.scale, x.shift, gvec, ctx.stats, coef_of())
            x.bwd = (gx, part, None, gx._version)
            return tag(gx)
        if g is not None and rows16 and g.shape == y.shape and g.is_contiguous() and ops.conv3x3_dgrad_applyfused_supported(64, rows, dtype):
            wpt = _packed(conv, 64, rows, dtype, None, True)
            if feed_stats and rows == 64:
                _, gx, part = ops.conv3x3_dgrad_applyfused(g, y, ctx.stats, coef_of(), wpt, x.t, x.scale, x.shift, reverse=d, want_dy=False)
                x.bwd = (gx, part, None, gx._version)
            else:
                _, gx, _ = ops.conv3x3_dgrad_applyfused(g, y, ctx.stats, coef_of(), wpt, reverse=d, want_dy=False)
            return tag(gx)
        # shapes / dtypes without a fused input-gradient kernel: the apply pass, then the plain input gradient
        dy = ops.bn_bwd(g, gvec, y, ctx.stats, Cout, gam, dgam, dbet, accumulate, dbias, coef=coef_of())
        gx, _ = ops.conv3x3_fwd(dy, _packed(conv, y.shape[-1], rows, dtype, ctx.perm, True), None, None, None, want_stats=False)
        return tag(gx, False)
    # A globally pooled layer (the gradient wrt its ReLU output is one row per sample): both consumers of dy -- the weight
    # gradient and the input gradient -- form it from (gvec, y) while staging their tiles; no apply pass, no dy tensor.
    if (gvec is not None and need_input_grad and x.scale is not None and perm_dev is None and ctx.perm is None and rows == 64
            and x.t.shape[-1] == 64 and ctx.stats.is_contiguous() and ops.conv3x3_gvfused_supported(64, y.shape[-1], dtype)):
        coef = coef_of()
        wpt = _packed(conv, y.shape[-1], rows, dtype, None, True)
        if (feed_stats and y.shape[-1] == 64 and ops.conv3x3_bwd_fused_supported(dtype, y.shape) and gvec.shape[-1] == 64
                and y.shape[0] <= ops.conv3x3_bwd_fused_gvec_max_batch()):
            # one kernel for both gradients (csrc/bwd_ws.hip, gvec form): y and the feeding layer's raw output are read once
            gx, part, pcoef = ops.conv3x3_bwd_fused(None, y, ctx.stats, coef, wpt, x.t, x.scale, x.shift, grads[conv.weight], accumulate,
                                                    fin=rider, gvec=gvec)
            x.bwd = (gx, part, pcoef, gx._version)
            return tag(gx)
        if feed_stats:   # input gradient first: the sums it emits are finished by a rider on the weight gradient's reduction
            gx, part = ops.conv3x3_dgrad_bwdstats(y, wpt, x.t, x.scale, x.shift, gvec, ctx.stats, coef)
            pcoef = ops.conv3x3_wgrad_gvfused(x.t, x.scale, x.shift, gvec, y, ctx.stats, coef, grads[conv.weight], accumulate, fin=rider(part))
            x.bwd = (gx, part, pcoef, gx._version)
            return tag(gx)
        ops.conv3x3_wgrad_gvfused(x.t, x.scale, x.shift, gvec, y, ctx.stats, coef, grads[conv.weight], accumulate)
        return tag(ops.conv3x3_dgrad_gvfused(y, wpt, gvec, ctx.stats, coef))
    # An image-fed first layer whose input needs a gradient too (the decoder's and the discriminator's conv1): one kernel forms dy from
    # (g, y) once and feeds both the input gradient (16 channels, 3 real) and the weight gradient (csrc/bwd_ws16.hip); no dy tensor
    if (g is not None and need_input_grad and x.scale is None and perm_dev is None and ctx.perm is None and dgrad_channels is None
            and x.t.shape[-1] == 16 and y.shape[-1] == 64 and g.shape == y.shape and g.is_contiguous() and ctx.stats.is_contiguous()
            and conv.weight.shape[1] <= 16 and ops.conv3x3_bwd_fused16_supported(y.shape, dtype)):
        gx = ops.conv3x3_bwd_fused16(g, y, ctx.stats, coef_of(), _packed(conv, 64, 16, dtype, None, True), x.t, grads[conv.weight], accumulate,
                                     reverse=d, premasked=pre is not None and getattr(g, "_wm_masked", False))
        return tag(gx)
    # An ordinary 64 -> 64 layer: the input-gradient kernel reads g and y, forms dy while staging and leaves it in memory
    # for the weight gradient -- the stand-alone apply pass is gone.
    if (g is not None and need_input_grad and perm_dev is None and ctx.perm is None and rows in (64, 32)
            and y.shape[-1] == 64 and g.shape == y.shape and g.is_contiguous() and ctx.stats.is_contiguous()
            and ops.conv3x3_dgrad_applyfused_supported(64, rows, dtype)):
        coef = coef_of()
        wpt = _packed(conv, 64, rows, dtype, None, True)
        if feed_stats and rows == 64 and ops.conv3x3_bwd_fused_supported(dtype, y.shape):
            # one kernel: dy is formed in the LDS and feeds BOTH gradients; the feeding layer's raw output is read once (csrc/bwd_ws.hip)
            # (g itself may come from such a kernel, which writes its gradient already multiplied by this layer's ReLU mask: the staging
            # then skips the mask arithmetic -- trusted only for exactly that tensor, unmodified: the check in coef_of())
            gx, part, pcoef = ops.conv3x3_bwd_fused(g, y, ctx.stats, coef, wpt, x.t, x.scale, x.shift, grads[conv.weight], accumulate,
                                                    reverse=d, fin=rider, premasked=pre is not None and getattr(g, "_wm_masked", False))
            x.bwd = (gx, part, pcoef, gx._version)
        elif feed_stats:
            dy, gx, part = ops.conv3x3_dgrad_applyfused(g, y, ctx.stats, coef, wpt, x.t, x.scale, x.shift, reverse=d)
            pcoef = ops.conv3x3_wgrad(x.t, x.t.shape[-1], x.scale, x.shift, dy, grads[conv.weight], accumulate, reverse=not d, fin=rider(part))
            x.bwd = (gx, part, pcoef, gx._version)
        else:
            dy, gx, _ = ops.conv3x3_dgrad_applyfused(g, y, ctx.stats, coef, wpt, reverse=d)
            ops.conv3x3_wgrad(x.t, x.t.shape[-1], x.scale, x.shift, dy, grads[conv.weight], accumulate, reverse=not d)
        return tag(gx)
    dy = ops.bn_bwd(g, gvec, y, ctx.stats, Cout, gam, dgam, dbet, accumulate, dbias, coef=coef_of())
    ops.conv3x3_wgrad(x.t, x.t.shape[-1], x.scale, x.shift, dy, grads[conv.weight], accumulate, perm_dev=perm_dev, reverse=True)   # the apply pass swept forwards
    if not need_input_grad:
        return None
    wpt = _packed(conv, y.shape[-1], rows, dtype, ctx.perm, True)
    if feed_stats:
        gx, part = ops.conv3x3_dgrad_bwdstats(dy, wpt, x.t, x.scale, x.shift)
        x.bwd = (gx, part, None, gx._version)
        return tag(gx, False)
    gx, _ = ops.conv3x3_fwd(dy, wpt, None, None, None, want_stats=False)
    return tag(gx, False)


def grad_dict(module, flat_grad=None):
    """param -> gradient view.  With flat_grad None, uses/creates p.grad."""
    out = {}
    off = 0
    for p in module.parameters():
        if flat_grad is None:
            if p.grad is None:
                p.grad = torch.zeros_like(p.data)
            out[p] = p.grad
        else:
            n = p.numel()
            out[p] = flat_grad[off:off + n].view_as(p)
            off += n
    return out


class FlatModule:
    """Mixin: keeps all parameters of an nn.Module in one flat f32 buffer (and the gradients in
    another) so the optimiser step is one kernel and the gradient all-reduce one RCCL bucket."""

    def flatten_parameters_(self):
        params = list(self.parameters())
        if not params:
            return
        dev = params[0].device
        n = sum(p.numel() for p in params)
        flat = getattr(self, "_flat", None)
        ok = flat is not None and flat.device == dev and flat.numel() == n
        if ok:
            off = 0
            for p in params:
                if p.data_ptr() != flat.data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if ok:
            return
        flat = torch.empty(n, device=dev, dtype=torch.float32)
        gflat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            k = p.numel()
            v = flat[off:off + k].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = gflat[off:off + k].view_as(p)
            off += k
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_gflat", gflat)
        if not getattr(self, "_wm_load_hook", False):
            # a state_dict load writes the parameters behind every version check (copy_ into `.data` aliases): drop the packs
            self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packs())
            object.__setattr__(self, "_wm_load_hook", True)
        # BatchNorm step counters: views of one int64 tensor, so a forward bumps them with one launch
        bns = [m for m in self.modules() if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None]
        if bns:
            nbt = torch.stack([m.num_batches_tracked.to(dev) for m in bns])
            for i, m in enumerate(bns):
                m._buffers["num_batches_tracked"] = nbt[i]
            object.__setattr__(self, "_nbt", nbt)

    # ---- packed conv weights: one launch per optimiser step instead of one per conv call
    def pack_plan(self):
        """the network's ops.PackPlan (created on first use; every 3x3 conv of the module points at it)."""
        plan = getattr(self, "_pack_plan", None)
        flat = getattr(self, "_flat", None)
        if plan is None or getattr(self, "_pack_plan_flat", None) is not flat:
            plan = ops.PackPlan()
            object.__setattr__(self, "_pack_plan", plan)
            object.__setattr__(self, "_pack_plan_flat", flat)   # a re-flatten moves the parameters: start over
            for m in self.modules():
                if isinstance(m, torch.nn.Conv2d) and tuple(m.kernel_size) == (3, 3):
                    object.__setattr__(m, "_wm_plan", plan)
        return plan

    def refresh_packs(self):
        """(re)pack every conv weight the network has used so far from the CURRENT parameters, in one launch; the packs
        stay valid until invalidate_packs().  Callers own the validity window (Hidden.train_on_batch)."""
        self.flatten_parameters_()
        self.pack_plan().refresh()

    def invalidate_packs(self):
        plan = getattr(self, "_pack_plan", None)
        if plan is not None:
            plan.invalidate()

    @property
    def flat_params(self):
        self.flatten_parameters_()
        return self._flat

    @property
    def flat_grads(self):
        self.flatten_parameters_()
        return self._gflat


def set_compute_dtype(module, dtype):
    """torch.bfloat16 (production), torch.float16 (the reference's autocast dtype; needs loss scaling, see models/IRNrhi_model.py)
    or torch.float32 (parity path) for every HIP-backed submodule."""
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError("compute dtype must be torch.float32, torch.bfloat16 or torch.float16")
    for m in module.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dtype
    return module


@contextlib.contextmanager
def defer_bn_counters():
    """inside the block, bump_bn_counters only counts; one multi-tensor add applies everything at exit (a training step
    runs five network forwards: one launch instead of five)"""
    if getattr(_tls, "bumps", None) is not None:   # nested: the outer block flushes
        yield
        return
    _tls.bumps = {}
    try:
        yield
    finally:
        pend, _tls.bumps = _tls.bumps, None
        if pend:
            torch._foreach_add_([m._nbt for m in pend], [pend[m] for m in pend])


def bump_bn_counters(module):
    """num_batches_tracked += 1 for every BatchNorm2d of `module` (one launch when the counters are flat)."""
    nbt = getattr(module, "_nbt", None)
    if nbt is not None and all(m.num_batches_tracked.data_ptr() == nbt[i].data_ptr() for i, m in enumerate(
            mm for mm in module.modules() if isinstance(mm, torch.nn.BatchNorm2d) and mm.num_batches_tracked is not None)):
        pend = getattr(_tls, "bumps", None)
        if pend is not None:
            pend[module] = pend.get(module, 0) + 1
            return
        nbt += 1
        return
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None:
            m.num_batches_tracked += 1
