from .ms_deform_attn import MSDeformAttnIDOL, MSDeformAttnSeqFormer  # noqa: F401
