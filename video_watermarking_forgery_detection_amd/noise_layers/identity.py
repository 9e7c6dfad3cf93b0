"""Identity noise layer (reference noise_layers/identity.py:5-16)."""
import torch.nn as nn


class Identity(nn.Module):
    capturable = True    # deterministic launches: Hidden.enable_graph may capture a step through this layer

    def __init__(self):
        super(Identity, self).__init__()
        self.name = "Identity"

    def forward(self, image):
        return image

    # explicit (autograd-free) interface used by the training step
    def fwd(self, image):
        return image, None

    def bwd(self, ctx, g):
        return g
