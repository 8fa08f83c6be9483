"""Drop-in for the reference's `_gridencoder` pybind module: put this directory on
PYTHONPATH and `external/gridencoder/grid.py:9-12` picks the HIP backend up unchanged."""
from sparsefusion_amd.gridencoder.backend import grid_encode_forward, grid_encode_backward  # noqa: F401
