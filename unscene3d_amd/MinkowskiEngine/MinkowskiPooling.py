"""`from MinkowskiEngine.MinkowskiPooling import MinkowskiAvgPooling` (reference models/mask3d.py:5)."""
from . import MinkowskiAvgPooling  # noqa: F401
