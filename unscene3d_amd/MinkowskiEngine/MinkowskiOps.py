"""`import MinkowskiEngine.MinkowskiOps as me` surface used by the reference
(models/res16unet.py:1, models/mask3d.py:4): `me.cat`, `me.SparseTensor`."""
from . import SparseTensor, cat  # noqa: F401
