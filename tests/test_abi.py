"""CPU: libvnext_hip.so loads without a GPU and exports exactly what include/*.h
declares; the ctypes table in vnext_amd/_lib.py covers every declaration.  No
compute entry point is called here."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(vnx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


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
    # the drop-in boundary itself is vnext_hip.h; the debug header only adds vnx_debug_* names
    assert all(n.startswith("vnx_debug_") for n in _lib.DEBUG_SIGNATURES)
    assert not any(n.startswith("vnx_debug_") for n in _lib.SIGNATURES)


def test_library_exports_nothing_undeclared():
    """The other direction: every C symbol the product library exports is declared in include/."""
    import subprocess
    from vnext_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines()
                      if len(line.split()) == 3 and line.split()[1] == "T" and not line.split()[2].startswith("_"))
    assert exported == declared_functions()


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
