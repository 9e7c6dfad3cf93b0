"""EncoderDecoder -- mirror of the reference's hidden_models/encoder_decoder.py:8-29:
encoder -> noiser([encoded, cover])[0] -> decoder, returning (encoded, noised, decoded)."""
import torch.nn as nn

from ..options import HiDDenConfiguration
from .decoder import Decoder
from .encoder import Encoder


class EncoderDecoder(nn.Module):
    def __init__(self, config: HiDDenConfiguration, noiser):
        super(EncoderDecoder, self).__init__()
        self.encoder = Encoder(config)
        self.noiser = noiser
        self.decoder = Decoder(config)

    def forward(self, image, message):
        encoded_image = self.encoder(image, message)
        noised_and_cover = self.noiser([encoded_image, image])
        noised_image = noised_and_cover[0]
        decoded_message = self.decoder(noised_image)
        return encoded_image, noised_image, decoded_message
