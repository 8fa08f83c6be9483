from .encoder import GridEncoder, grid_encode  # noqa: F401
