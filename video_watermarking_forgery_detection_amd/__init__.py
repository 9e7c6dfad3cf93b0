"""MI355X-native watermark embed -> attack -> extract training step.

Host side (Python, mirrors the reference's module/class names):
    options.HiDDenConfiguration
    hidden_models.{ConvBNRelu, Encoder, Decoder, Discriminator, EncoderDecoder, Hidden}
    noise_layers.{Jpeg, JpegSS, JpegMask, Combined, Identity, Noiser, ...}
Device side: hand-written gfx950 HIP kernels in csrc/, reached only through the C ABI of
include/wm_hip.h (lib/libwm_hip.so, loaded with ctypes).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .engine import set_compute_dtype  # noqa: F401
from .options import HiDDenConfiguration  # noqa: F401

__version__ = "0.1.0"
