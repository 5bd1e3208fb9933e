"""vnext_amd -- MI355X-native hot path for VNext (SeqFormer / IDOL).

Host side of libvnext_hip.so (hand-written HIP for gfx950): the
`MultiScaleDeformableAttention` extension surface, `MSDeformAttnFunction`, the
`MSDeformAttn` modules of both projects, and the heads built on them.
"""
__version__ = "0.1.0"
