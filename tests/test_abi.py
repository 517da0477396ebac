"""The C-ABI library builds for gfx950 (cross-compile, no GPU), loads, and exports every symbol declared in
include/vse_hip.h; record layouts agree between ir.py and the C structs.  CPU only — no compute calls."""
import ctypes
import os
import re

from vse_amd import engine, ir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vse_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vse_[a-z_]+)\s*\(", src)))


def test_library_exports_header(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vse_hip.h but not exported"
    assert set(names) == set(engine.EXPORTS)


def test_record_layouts(built_lib):
    lib = engine.load_library()
    assert lib.vse_sizeof_op() == ir.OP_DT.itemsize == 352
    assert lib.vse_sizeof_view() == ir.VIEW_DT.itemsize == 40
    assert lib.vse_abi_version() == 2
    assert lib.vse_is_dev_build() == 0            # the in-tree library is the product build: no experiment switch reaches it


def test_product_refuses_without_gpu(built_lib):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.VseError):
        engine.Context(0)
