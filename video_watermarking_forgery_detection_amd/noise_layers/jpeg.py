"""Jpeg / JpegSS / JpegMask -- mirror of the reference's noise_layers/jpeg.py:48-306 on the fused
block-JPEG HIP kernels (csrc/jpeg.hip): one launch forward, one backward, 24 B/px each.

Same class names, ctor arguments (`Q`, `subsample=0`), `.name` ("Jpeg50", "JpegSS50", "JpegMask50"),
`scale_factor` formula (jpeg.py:221) and no output clamp (jpeg.py:240).  The quantisation tables are
built on the host exactly like std_quantization does -- (table * scale_factor).round().clamp(min=1) in
fp32 (jpeg.py:54-76) -- and handed to the kernel by value.
"""
import torch
import torch.nn as nn

from .. import ops

_LUM = [
    [16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
    [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
    [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]]
_CHROMA = [
    [17, 18, 24, 47, 99, 99, 99, 99], [18, 21, 26, 66, 99, 99, 99, 99], [24, 26, 56, 99, 99, 99, 99, 99],
    [47, 66, 99, 99, 99, 99, 99, 99], [99, 99, 99, 99, 99, 99, 99, 99], [99, 99, 99, 99, 99, 99, 99, 99],
    [99, 99, 99, 99, 99, 99, 99, 99], [99, 99, 99, 99, 99, 99, 99, 99]]


class _JpegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, layer):
        x = image.float().contiguous()
        ctx.layer = layer
        ctx.save_for_backward(x if layer._mode == ops.JPEG_SS else None)
        return ops.jpeg_fwd(x, layer._mode, layer._tables, layer.subsample)

    @staticmethod
    def backward(ctx, g):
        layer = ctx.layer
        (x,) = ctx.saved_tensors
        return ops.jpeg_bwd(x, g.float().contiguous(), layer._mode, layer._tables, layer.subsample), None


class JpegBasic(nn.Module):
    _mode = ops.JPEG_ROUND
    _prefix = "Jpeg"
    capturable = True    # fwd / bwd launch the same kernels with the same arguments every call (no host-side randomness): a training step
                         # through this layer may be captured into a hipGraph (Hidden.enable_graph)

    def __init__(self, Q=50, subsample=0):
        super(JpegBasic, self).__init__()
        self.name = self._prefix + str(Q)
        self.Q = Q
        self.scale_factor = 2 - self.Q * 0.02 if self.Q >= 50 else 50 / self.Q
        self.subsample = subsample
        if subsample not in (0, 2):
            raise ValueError("subsample must be 0 or 2")
        lum = (torch.tensor(_LUM, dtype=torch.float) * self.scale_factor).round().clamp(min=1)
        chroma = (torch.tensor(_CHROMA, dtype=torch.float) * self.scale_factor).round().clamp(min=1)
        self._tables = lum.flatten().tolist() + chroma.flatten().tolist()

    def forward(self, image):
        if not image.is_cuda:
            raise RuntimeError(self.name + " runs on the HIP path only: move the input to cuda")
        return _JpegFn.apply(image, self)

    # explicit (autograd-free) interface used by the training step
    def fwd(self, image):
        from .. import engine
        x = image.contiguous()
        dt = engine.wanted_image_act_dtype()
        if dt is not None:   # inside a training step: the decoder's NHWC16 input written by this kernel too (no conversion launch)
            y, a16 = ops.jpeg_fwd(x, self._mode, self._tables, self.subsample, act16_dtype=dt)
            engine.register_image_act(y, a16)
        else:
            y = ops.jpeg_fwd(x, self._mode, self._tables, self.subsample)
        return y, (x if self._mode == ops.JPEG_SS else None)

    def bwd(self, ctx, g):
        return ops.jpeg_bwd(ctx, g, self._mode, self._tables, self.subsample)

    def bwd_is_zero(self, ctx):
        """is the gradient this layer passes back identically zero?  Jpeg: yes -- torch.round has zero gradient (jpeg.py:226-240 of the
        reference: autograd multiplies by 0 at every coefficient), so bwd() returns zeros whatever g is.  A training step may then skip
        producing g (the decoder's gradient wrt its input) and the addition of this layer's zeros (hidden_models/hidden.py)"""
        return self._mode == ops.JPEG_ROUND


class Jpeg(JpegBasic):
    """torch.round quantisation: zero gradient, like the reference."""
    _mode = ops.JPEG_ROUND
    _prefix = "Jpeg"


class JpegSS(JpegBasic):
    """round_ss: x^3 inside |x| < 0.5, identity outside (jpeg.py:255-257)."""
    _mode = ops.JPEG_SS
    _prefix = "JpegSS"


class JpegMask(JpegBasic):
    """no quantisation: keep the 5x5 low-frequency Y and 3x3 U/V coefficients (jpeg.py:288-291)."""
    _mode = ops.JPEG_MASK
    _prefix = "JpegMask"
