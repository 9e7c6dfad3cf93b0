"""Combined -- random (or `id`-selected) choice among child attack layers; mirror of the
reference's noise_layers/combined.py:6-20 (python `random.randint`, `.name` of the chosen child)."""
import torch.nn as nn

from . import get_random_int
from .identity import Identity


class Combined(nn.Module):
    def __init__(self, list=None):
        super(Combined, self).__init__()
        if list is None:
            list = [Identity()]
        self.list = nn.ModuleList(list) if all(isinstance(m, nn.Module) for m in list) else list
        self.name = "NotChosenYet"

    def _pick(self, id):
        if id is None or id >= len(self.list):
            id = get_random_int([0, len(self.list) - 1])
        selected = self.list[id]
        self.name = selected.name
        return selected

    def forward(self, image_and_cover, id=None):
        return self._pick(id)(image_and_cover)

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
