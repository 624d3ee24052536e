"""unscene3d_amd — MI355X-native hot path of UnScene3D.

Importing the package loads libusc3d_hip.so (ImportError if it was not built —
there is no CPU fallback)."""
from . import _lib  # noqa: F401  (fails loudly when the HIP extension is missing)

__all__ = ["ops", "MinkowskiEngine", "models"]
