"""ctypes binding of libsparsefusion_hip.so (C ABI declared in include/sparsefusion_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or a tensor
is not on a GPU, the wrappers raise.  torch is used only for device memory and
streams (``tensor.data_ptr()``, ``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_HIP_LIB: an A/B build of the SAME library (sparsefusion_amd/build.py::build_variant), for tuning runs on the GPU box
# SF_OPERAND=f16: the build with IEEE-half MFMA operands (csrc/sf_operand.h; build.build_f16) instead of bf16, for the whole process
_DEFAULT_LIB = "libsparsefusion_hip_f16.so" if os.environ.get("SF_OPERAND", "bf16").lower() in ("f16", "fp16", "half") else "libsparsefusion_hip.so"
LIB_PATH = os.environ.get("SF_HIP_LIB") or os.path.join(_HERE, _DEFAULT_LIB)

_lib = None

c_f32p = C.c_void_p   # device pointers travel as integers
c_i32p = C.c_void_p
u32 = C.c_uint32
u64 = C.c_uint64


class SfNgpField(C.Structure):
    _fields_ = [("embeddings", C.c_void_p), ("h_offsets", C.c_void_p), ("L", u32), ("S", C.c_float),
                ("H", u32), ("gridtype", u32),
                ("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("bound", C.c_float)]


class SfNgpFieldGrad(C.Structure):
    _fields_ = [("g_embeddings", C.c_void_p), ("g_w0", C.c_void_p), ("g_b0", C.c_void_p),
                ("g_w1", C.c_void_p), ("g_b1", C.c_void_p), ("g_w2", C.c_void_p), ("g_b2", C.c_void_p)]


SF_ADAM_MAX_TENSORS = 16


class SfAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_uint64), ("step_size", C.c_float), ("pad_", C.c_float)]


class SfAdamArgs(C.Structure):
    _fields_ = [("t", SfAdamTensor * SF_ADAM_MAX_TENSORS), ("chunk_start", C.c_uint32 * SF_ADAM_MAX_TENSORS),
                ("n_tensors", C.c_uint32), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias_correction2_sqrt", C.c_float), ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float)]


class SfOp(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_int32), ("p", C.c_void_p * 24),
                ("i", C.c_int32 * 32), ("f", C.c_float * 8)]


# name -> (restype, argtypes); mirrors include/sparsefusion_hip.h one to one.
SIGNATURES = {
    "sf_last_error": (C.c_char_p, []),
    "sf_abi_version": (C.c_int, []),
    "sf_operand_is_f16": (C.c_int, []),
    "sf_grid_encode_forward": (C.c_int, [c_f32p, c_f32p, c_i32p, c_f32p, u32, u32, u32, u32, C.c_float, u32,
                                         c_f32p, u32, C.c_int, C.c_void_p, C.c_void_p]),
    "sf_grid_encode_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, u32, u32, u32, u32, C.c_float,
                                          u32, c_f32p, c_f32p, u32, C.c_int, C.c_void_p, C.c_void_p]),
    "sf_near_far_from_aabb": (C.c_int, [c_f32p, c_f32p, c_f32p, u32, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    "sf_morton3D": (C.c_int, [c_i32p, u32, c_i32p, C.c_void_p]),
    "sf_morton3D_invert": (C.c_int, [c_i32p, u32, c_i32p, C.c_void_p]),
    "sf_packbits": (C.c_int, [c_f32p, u32, C.c_float, C.c_void_p, C.c_void_p]),
    "sf_sph_from_ray": (C.c_int, [c_f32p, c_f32p, C.c_float, u32, c_f32p, C.c_void_p]),
    "sf_march_rays_train_workspace_bytes": (u64, [u32]),
    "sf_march_rays_train": (C.c_int, [c_f32p, c_f32p, C.c_void_p, C.c_float, C.c_float, u32, u32, u32, u32, u32, c_f32p,
                                      c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_f32p, C.c_void_p, u64, C.c_void_p]),
    "sf_composite_rays_train_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, u32, u32, C.c_float, c_f32p, c_f32p,
                                                  c_f32p, C.c_void_p]),
    "sf_composite_rays_train_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, u32,
                                                   u32, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    "sf_march_rays": (C.c_int, [u32, u32, c_i32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_float, u32, u32, u32,
                                C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sf_composite_rays": (C.c_int, [u32, u32, C.c_float, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                    c_f32p, C.c_void_p]),
    "sf_adam_multi": (C.c_int, [C.POINTER(SfAdamArgs), C.c_void_p]),
    "sf_loss_partial_rows": (u32, []),
    "sf_upsample2x_forward": (C.c_int, [c_f32p, c_f32p, u32, u32, u32, C.c_void_p]),
    "sf_upsample2x_backward": (C.c_int, [c_f32p, c_f32p, u32, u32, u32, C.c_void_p]),
    "sf_render_loss_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, u64, u64, C.c_float, c_f32p, C.c_void_p]),
    "sf_render_loss_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, u64, u64, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sf_fusion_loss_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, u32, u64, u64, c_f32p, C.c_void_p]),
    "sf_fusion_loss_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, u32, u64, u64, C.c_float, C.c_float, C.c_float,
                                          c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sf_ngp_density": (C.c_int, [C.POINTER(SfNgpField), c_f32p, u32, c_f32p, c_f32p, C.c_void_p]),
    "sf_ngp_render_forward": (C.c_int, [C.POINTER(SfNgpField), c_f32p, c_f32p, c_f32p, u32, u32, C.c_float,
                                        c_f32p, c_f32p, c_f32p, u32, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p,
                                        c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, u64, C.c_void_p]),
    "sf_ngp_render_backward": (C.c_int, [C.POINTER(SfNgpField), C.POINTER(SfNgpFieldGrad), c_f32p, c_f32p,
                                         c_f32p, u32, u32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                         C.c_float, c_f32p, c_f32p, u32, c_f32p, c_f32p, u64, C.c_void_p]),
    "sf_ngp_render_workspace_bytes": (u64, [u32, u32]),
    "sf_ngp_render_forward_workspace_bytes": (u64, [u32, u32]),
    "sf_ngp_render_cache_bytes": (u64, [u32, u32]),
    "sf_ngp_render_occ_eval": (C.c_int, [C.POINTER(SfNgpField), c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_float, u32, u32, u32,
                                         c_f32p, C.c_float, u32, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "sf_plan_run": (C.c_int, [C.POINTER(SfOp), u32, C.c_void_p]),
    "sf_plan_profile": (C.c_int, [C.POINTER(SfOp), u32, C.c_void_p, C.c_void_p]),
    "sf_conv_packed_elems": (u64, [u32, u32, u32, u32]),
    "sf_conv_pack_weights": (C.c_int, [C.c_void_p, u32, u32, u32, u32, u32, C.c_void_p]),
    "sf_plms_update": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, u64, c_f32p, c_f32p, C.c_void_p]),
    "sf_plms_combine": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, u64, c_f32p, c_f32p, C.c_void_p]),
    "sf_plms_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_void_p, u64, c_f32p, c_f32p, c_f32p, u64, C.c_void_p]),
}


_handles = {}


def _load(path):
    if path not in _handles:
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m sparsefusion_amd.build` (`--f16` for the IEEE-half operand build; "
                "there is no CPU fallback for the sparsefusion_amd hot path)")
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _handles[path] = handle
    return _handles[path]


def lib(operand=None):
    """The HIP library (loaded once).  operand=None: the process default (LIB_PATH: bf16 operands unless SF_OPERAND=f16);
    "f16" / "bf16": that operand build explicitly -- a module (Unet.half()) may run on IEEE-half operands while the rest of the
    process stays on bf16.  Raises if the library has not been built."""
    global _lib
    if operand is None:
        if _lib is None:
            _lib = _load(LIB_PATH)
        return _lib
    if operand == "f16":
        return _load(os.path.join(_HERE, "libsparsefusion_hip_f16.so"))
    if operand == "bf16":
        return _load(os.path.join(_HERE, "libsparsefusion_hip.so"))
    raise ValueError(f"unknown MFMA operand type {operand!r} (bf16 | f16)")


def operand_dtype(handle=None):
    """torch dtype of the MFMA operands of a library handle (default: the process library): bfloat16 or float16.  Host-side
    weight tables that are handed to the library as raw 16-bit values must be rounded to THIS type."""
    return torch.float16 if (handle or lib()).sf_operand_is_f16() else torch.bfloat16


def check(rc, what="", handle=None):
    if rc != 0:
        msg = (handle or lib()).sf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg}" if what else msg)


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sparsefusion_amd: tensor must live on a HIP device (no CPU path)")


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
