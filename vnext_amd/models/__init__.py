"""Importing this package registers the meta-architectures on the META_ARCH_REGISTRY surface."""
from .seqformer import SeqFormer  # noqa: F401
from .idol import IDOL  # noqa: F401
