from .dynamic_mask import dynamic_mask_head, dynamic_mask_with_coords  # noqa: F401
from .reid import bisoftmax, loss_reid, match_scores, similarity  # noqa: F401
