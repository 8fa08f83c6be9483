"""`_gridencoder` backend: the two entry points of the reference's pybind module
(external/gridencoder/src/bindings.cpp:6-7) with the SAME positional signatures,
implemented on libsparsefusion_hip.so.  `external/gridencoder/grid.py:9-12` does
`import _gridencoder as _backend`; sparsefusion_amd/shims/_gridencoder.py re-exports
this module under that name.

Outputs are caller-allocated and filled in place (grid.py:42-47, :72-77)."""
import numpy as np
import torch

from .. import _lib


def _host_offsets(offsets):
    """Host copy of the (tiny) offsets tensor, cached on the tensor object so the
    launch needs no device->host sync after the first call."""
    cached = getattr(offsets, "_sf_host", None)
    if cached is None or cached[0] != offsets.data_ptr() or cached[1] != offsets._version:
        host = np.ascontiguousarray(offsets.detach().cpu().numpy().astype(np.int32))
        cached = (offsets.data_ptr(), offsets._version, host)
        try:
            offsets._sf_host = cached
        except Exception:
            pass
    return cached[2]


def _check(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor (fp32 path only; the reference loop is fp32)")


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners):
    _check(inputs, "inputs"); _check(embeddings, "embeddings"); _check(outputs, "outputs")
    _check(offsets, "offsets", torch.int32)
    if dy_dx is not None:
        _check(dy_dx, "dy_dx")
    ho = _host_offsets(offsets)
    rc = _lib.lib().sf_grid_encode_forward(
        _lib.ptr(inputs), _lib.ptr(embeddings), _lib.ptr(offsets), _lib.ptr(outputs),
        int(B), int(D), int(C), int(L), float(S), int(H), _lib.ptr(dy_dx), int(gridtype),
        int(bool(align_corners)), ho.ctypes.data, _lib.stream_ptr())
    _lib.check(rc, "grid_encode_forward")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners):
    _check(grad, "grad"); _check(inputs, "inputs"); _check(embeddings, "embeddings")
    _check(grad_embeddings, "grad_embeddings"); _check(offsets, "offsets", torch.int32)
    ho = _host_offsets(offsets)
    rc = _lib.lib().sf_grid_encode_backward(
        _lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings), _lib.ptr(offsets), _lib.ptr(grad_embeddings),
        int(B), int(D), int(C), int(L), float(S), int(H), _lib.ptr(dy_dx), _lib.ptr(grad_inputs),
        int(gridtype), int(bool(align_corners)), ho.ctypes.data, _lib.stream_ptr())
    _lib.check(rc, "grid_encode_backward")
