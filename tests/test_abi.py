"""CPU: the C-ABI library loads and exports every symbol include/abr_iod_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "abr_iod_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(abr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from abr_iod_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/abr_iod_hip.h but not exported"
    # and the Python binding table covers the same set
    assert set(_lib.EXPORTS) == set(names), set(_lib.EXPORTS) ^ set(names)


def test_version_and_error_channel():
    from abr_iod_amd import _lib
    L = _lib.lib()
    assert L.abr_version() >= 100
    # invalid argument -> negative status + message, no launch (works without a GPU)
    rc = L.abr_roi_align_forward(None, None, 4, 1, 8, 4, 4, 1.0, 0, 7, 0, 1, 1, None, None)
    assert rc == -1 and b"roi_align_forward" in L.abr_last_error()
    assert L.abr_nms_workspace_bytes(2, 12000) == 2 * 12000 * 188 * 8


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from abr_iod_amd import _C
    with pytest.raises(RuntimeError):
        _C.roi_align_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 1.0, 7, 7, 0)
    # nothing under abr_iod_amd/ may import the oracle
    for dp, _, fs in os.walk(os.path.join(ROOT, "abr_iod_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle\b", src, flags=re.M), f
                assert "liboracle" not in src and "oracle/" not in re.sub(r"#.*|\"\"\".*?\"\"\"", "", src, flags=re.S), f
