"""Base classes (reference models/model.py:4-17)."""
from ..MinkowskiEngine import MinkowskiNetwork


class Model(MinkowskiNetwork):
    """Base of every sparse convnet: remembers channels/config, D-dimensional."""

    OUT_PIXEL_DIST = -1

    def __init__(self, in_channels, out_channels, config, D, **kwargs):
        super().__init__(D)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.config = config
