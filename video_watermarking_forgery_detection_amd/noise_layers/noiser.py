"""Noiser -- the reference imports `noise_layers.noiser.Noiser` (hidden_models/encoder_decoder.py:5,
hidden.py:9) but the file is absent from its tree.  Contract taken from the call site
(encoder_decoder.py:26-27): called with [encoded_image, cover_image], returns a list whose first
element is the noised image.  One layer is drawn per call (upstream HiDDeN behaviour)."""
import random

import torch.nn as nn

from .identity import Identity


class Noiser(nn.Module):
    def __init__(self, noise_layers=None, device=None):
        super(Noiser, self).__init__()
        layers = list(noise_layers) if noise_layers else [Identity()]
        self.noise_layers = nn.ModuleList(layers)
        self.last = None

    def _pick(self, id=None):
        if id is None or id >= len(self.noise_layers):
            id = random.randint(0, len(self.noise_layers) - 1)
        self.last = self.noise_layers[id]
        return self.last

    def forward(self, encoded_and_cover, id=None):
        enc, cover = encoded_and_cover
        out = self._pick(id)(enc)
        if isinstance(out, tuple):  # Crop returns (image, apex)
            out = out[0]
        return [out, cover]

    def fwd(self, image, id=None):
        sel = self._pick(id)
        y, c = sel.fwd(image)
        return y, (sel, c)

    def bwd(self, ctx, g):
        sel, c = ctx
        return sel.bwd(c, g)

    def bwd_is_zero(self, ctx):
        sel, c = ctx
        return bool(getattr(sel, "bwd_is_zero", lambda _c: False)(c))
