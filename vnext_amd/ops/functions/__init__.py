from .ms_deform_attn_func import (MSDeformAttnFunction, MSDeformAttnFusedFunction, check_flattened_length,  # noqa: F401
                                   level_tensors, mark_levels_packed, ms_deform_attn)
