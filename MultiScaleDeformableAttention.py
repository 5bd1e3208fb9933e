"""Drop-in for the reference's compiled extension module of the same name
(built by projects/*/models/ops/setup.py:53,63 and imported at
projects/*/models/ops/functions/ms_deform_attn_func.py:18).  With this
directory on sys.path the reference's own `MSDeformAttnFunction` resolves to the
MI355X library unchanged.  Implementation: vnext_amd/msda_ext.py ->
libvnext_hip.so (include/vnext_hip.h).
"""
from vnext_amd.msda_ext import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
