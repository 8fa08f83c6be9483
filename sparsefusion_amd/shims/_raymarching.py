"""Drop-in for the reference's `_raymarching` pybind module
(raymarching/raymarching.py:9-12 does `import _raymarching as _backend`)."""
from sparsefusion_amd.raymarching.backend import (  # noqa: F401
    near_far_from_aabb, morton3D, morton3D_invert, packbits)
