"""Drop-in for the reference's `_raymarching` pybind module
(raymarching/raymarching.py:9-12 does `import _raymarching as _backend`): all ten entry points of
raymarching/src/bindings.cpp:7-18 with their positional signatures."""
from sparsefusion_amd.raymarching.backend import (  # noqa: F401
    near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits, march_rays_train,
    composite_rays_train_forward, composite_rays_train_backward, march_rays, composite_rays)
