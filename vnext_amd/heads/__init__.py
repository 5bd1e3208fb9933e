from .dynamic_mask import dynamic_mask_with_coords  # noqa: F401
