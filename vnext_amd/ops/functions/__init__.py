from .ms_deform_attn_func import MSDeformAttnFunction, mark_levels_packed, ms_deform_attn  # noqa: F401
