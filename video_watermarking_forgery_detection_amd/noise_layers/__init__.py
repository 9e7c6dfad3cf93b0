"""Differentiable attack layers on the MI355X kernels -- mirror of the reference's noise_layers/
package for the layers the models instantiate."""
import random


def get_random_float(float_range):
    return random.random() * (float_range[1] - float_range[0]) + float_range[0]


def get_random_int(int_range):
    return random.randint(int_range[0], int_range[1])


from .identity import Identity  # noqa: E402
from .jpeg import Jpeg, JpegSS, JpegMask, JpegBasic  # noqa: E402
from .combined import Combined  # noqa: E402
from .gaussian_blur import GaussianBlur  # noqa: E402
from .middle_filter import MiddleBlur  # noqa: E402
from .resize import Resize  # noqa: E402
from .crop import Crop  # noqa: E402
from .noiser import Noiser  # noqa: E402
