"""CPU: the C-ABI library loads and exports every symbol include/usc3d.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "usc3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(usc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from unscene3d_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/usc3d.h but not exported"
    # and the ctypes signature table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_error_channel():
    from unscene3d_amd import _lib

    assert _lib.lib.usc_abi_version() >= 1
    # argument validation happens before any HIP call, so it works without a GPU
    rc = _lib.lib.usc_voxel_floor_f64(None, -1, 0.02, None, None)
    assert rc == -1
    assert "usc_voxel_floor_f64" in _lib.last_error()
    assert _lib.lib.usc_coordmap_capacity(1000) == 2048


def test_ops_fail_loudly_without_device():
    import pytest
    import torch

    from unscene3d_amd import ops

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ops.coordmap_build(torch.zeros((4, 4), dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ops.furthest_point_sample(torch.zeros((1, 8, 3)), 4)
