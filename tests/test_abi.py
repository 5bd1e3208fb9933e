"""CPU: libvnext_hip.so loads without a GPU and exports exactly what include/*.h
declares; the ctypes table in vnext_amd/_lib.py covers every declaration.  No
compute entry point is called here."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


PRODUCT_HEADERS = ("vnext_hip.h", "vnext_hip_debug.h")
DEV_HEADER = "vnext_hip_dev.h"      # what only the development build (libvnext_hip_dev.so) exports


def declared_functions(headers=PRODUCT_HEADERS):
    names = []
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(vnx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines()
                  if len(line.split()) == 3 and line.split()[1] == "T" and not line.split()[2].startswith("_"))


def test_every_header_is_accounted_for():
    assert sorted(os.path.basename(h) for h in glob.glob(os.path.join(ROOT, "include", "*.h"))) == \
        sorted(PRODUCT_HEADERS + (DEV_HEADER,))


def test_header_declares_the_hot_path_entry_points():
    names = declared_functions()
    for required in ("vnx_msda_forward", "vnx_msda_backward", "vnx_abi_version", "vnx_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_functions():
        assert hasattr(hip_lib, name), f"{name} declared in include/ but not exported"


def test_ctypes_table_matches_header():
    from vnext_amd import _lib
    assert sorted({**_lib.SIGNATURES, **_lib.DEBUG_SIGNATURES}) == declared_functions()
    assert sorted(_lib.DEV_SIGNATURES) == declared_functions((DEV_HEADER,))
    # the drop-in boundary itself is vnext_hip.h; the debug header only adds vnx_debug_* names
    assert all(n.startswith("vnx_debug_") for n in _lib.DEBUG_SIGNATURES)
    assert not any(n.startswith("vnx_debug_") for n in _lib.SIGNATURES)
    assert sorted(_lib.SIGNATURES) == declared_functions(("vnext_hip.h",))


def test_library_exports_nothing_undeclared():
    """The other direction: every C symbol the product library exports is declared in include/."""
    from vnext_amd import _lib
    assert exported(_lib.LIB_PATH) == declared_functions()


def test_product_library_has_no_kernel_variant_knob():
    """the A/B / ablation variants (some return wrong results by construction) are compiled only into the
    development build; the product library neither exports a setter nor reads a process-wide value."""
    from vnext_amd import _lib
    names = exported(_lib.LIB_PATH)
    assert not [n for n in names if "variant" in n]
    assert "vnx_debug_read_tile_stamps" not in names and "vnx_debug_read_rec_stamps" not in names
    out = open(_lib.LIB_PATH, "rb").read()
    assert b"g_kernel_variant" not in out                  # not even as a local symbol
    assert b"msda_fwd_tile" not in out                     # the archived LDS-staged forwards are not linked in


def test_development_library_exports_the_product_surface_plus_the_dev_header():
    from vnext_amd import _lib
    assert exported(_lib.DEV_LIB_PATH) == sorted(declared_functions() + declared_functions((DEV_HEADER,)))


def test_a_nonzero_variant_routes_through_the_development_library():
    from vnext_amd import _lib
    assert _lib.lib() is _lib.product_lib()
    _lib.set_kernel_variant(13)
    try:
        assert _lib.lib() is _lib.dev_lib() and _lib.dev_lib().vnx_get_kernel_variant() == 13
        assert _lib.lib() is not _lib.product_lib()
    finally:
        _lib.set_kernel_variant(0)
    assert _lib.lib() is _lib.product_lib() and _lib.dev_lib().vnx_get_kernel_variant() == 0


def test_abi_version_and_status_strings(hip_lib):
    from vnext_amd import _lib
    assert hip_lib.vnx_abi_version() == _lib.ABI_VERSION
    assert hip_lib.vnx_status_string(0) == b"ok"
    assert b"unknown" in hip_lib.vnx_status_string(99)


def test_missing_library_fails_loudly(monkeypatch):
    from vnext_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "vnext_amd", "lib", "nope.so"))
    with pytest.raises(_lib.VnextHipError, match="no CPU fallback"):
        _lib.lib()


def test_cpu_tensors_are_rejected_like_the_reference():
    import torch
    import MultiScaleDeformableAttention as MSDA
    v = torch.zeros(1, 4, 1, 2)
    shapes = torch.tensor([[2, 2]])
    lsi = torch.tensor([0])
    loc = torch.zeros(1, 1, 1, 1, 1, 2)
    attn = torch.zeros(1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, torch.zeros(1, 1, 2), 64)
