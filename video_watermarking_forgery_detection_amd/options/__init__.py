"""`options.HiDDenConfiguration` -- imported by the reference's hidden_models/*.py
(encoder.py:3, decoder.py:2, discriminator.py:2, hidden.py:5) but absent from its tree
(options/__init__.py is empty).  Field names are pinned by the reference's usage
(encoder.py:13-25, decoder.py:15-22, discriminator.py:12-18, hidden.py:27,99-100); the
defaults are upstream HiDDeN's: L=30, encoder 4x64, decoder 7x64,
discriminator 3x64, loss weights decoder 1.0 / encoder 0.7 / adversarial 1e-3.
"""
import dataclasses


@dataclasses.dataclass
class HiDDenConfiguration:
    H: int
    W: int
    message_length: int = 30
    encoder_blocks: int = 4
    encoder_channels: int = 64
    decoder_blocks: int = 7
    decoder_channels: int = 64
    use_discriminator: bool = True
    use_vgg: bool = False
    discriminator_blocks: int = 3
    discriminator_channels: int = 64
    decoder_loss: float = 1.0
    encoder_loss: float = 0.7
    adversarial_loss: float = 1e-3
    enable_fp16: bool = False
