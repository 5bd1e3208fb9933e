from .ms_deform_attn_func import (MSDeformAttnFunction, check_flattened_length, level_tensors,  # noqa: F401
                                   mark_levels_packed, ms_deform_attn)
