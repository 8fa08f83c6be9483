"""ctypes binding of oracle/_ref/libref_native.so = the REFERENCE's own gridencoder.cu / raymarching.cu compiled for the host
by oracle/build_ref.py.  TEST INFRASTRUCTURE ONLY: used to pin oracle/ngp_ref.c (tests/test_oracle_native_pin.py) and to
generate tests/golden/ngp_native.pt (tests/golden/make_golden_native.py); never imported by sparsefusion_amd/.

Functions take the positional arguments of the reference's pybind entry points (raymarching/src/bindings.cpp:7-18,
external/gridencoder/src/bindings.cpp:6-7) as CPU torch tensors / Python scalars and raise RuntimeError with the
reference's own TORCH_CHECK / runtime_error message on failure."""
import contextlib
import ctypes as C

import torch

from . import build_ref

_lib = None
_unfused = False
_libs = {}
# p = tensor or None (pointer), u = uint32, f = float, i = int (bool)
_SPEC = {
    "grid_encode_forward": "ppppuuuufupui",
    "grid_encode_backward": "pppppuuuufuppui",
    "near_far_from_aabb": "pppufpp",
    "sph_from_ray": "ppfup",
    "morton3D": "pup",
    "morton3D_invert": "pup",
    "packbits": "pufp",
    "march_rays_train": "pppffuuuuupppppppp",
    "composite_rays_train_forward": "ppppuufppp",
    "composite_rays_train_backward": "ppppppppuufpp",
    "march_rays": "uuppppffuuuppppppp",
    "composite_rays": "uufpppppppp",
}


def available():
    return build_ref.build() is not None and build_ref.build(unfused=True) is not None


def lib():
    key = _unfused
    if key not in _libs:
        path = build_ref.build(unfused=key)
        if path is None:
            raise RuntimeError("oracle/_ref/libref_native*.so is absent and /root/reference is not here to build it")
        _libs[key] = C.CDLL(path)
        _libs[key].ref_last_error.restype = C.c_char_p
    return _libs[key]


@contextlib.contextmanager
def unfused():
    """Inside this context the functions run on the -ffp-contract=off build of the reference sources."""
    global _unfused
    prev, _unfused = _unfused, True
    try:
        yield
    finally:
        _unfused = prev


def _conv(kind, v):
    if kind == "p":
        if v is None:
            return C.c_void_p(0)
        assert isinstance(v, torch.Tensor) and v.device.type == "cpu" and v.is_contiguous(), "CPU contiguous tensors only"
        assert v.dtype in (torch.float32, torch.int32, torch.uint8), v.dtype
        return C.c_void_p(v.data_ptr())
    if kind == "u":
        return C.c_uint32(int(v))
    if kind == "f":
        return C.c_float(float(v))
    return C.c_int(int(bool(v)))


def _make(name):
    spec = _SPEC[name]

    def fn(*args):
        assert len(args) == len(spec), (name, len(args), len(spec))
        if getattr(lib(), "ref_" + name)(*[_conv(k, a) for k, a in zip(spec, args)]):
            raise RuntimeError(lib().ref_last_error().decode())
    fn.__name__ = name
    return fn


for _n in _SPEC:
    globals()[_n] = _make(_n)
