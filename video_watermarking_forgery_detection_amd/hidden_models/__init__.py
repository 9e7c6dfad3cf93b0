"""HiDDeN encoder / decoder / discriminator on the MI355X kernels -- mirror of the reference's
hidden_models/ package (same module and class names)."""
from .conv_bn_relu import ConvBNRelu
from .encoder import Encoder
from .decoder import Decoder
from .discriminator import Discriminator
from .encoder_decoder import EncoderDecoder
from .hidden import Hidden
