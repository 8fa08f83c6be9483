"""ctypes harness for tests/hostemu/fused_emu.cpp: runs SF_OP_FCONV / SF_OP_SLOTS / SF_OP_GCA ops of the product's kernel
source on CPU threads (host pointers).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import torch

from sparsefusion_amd import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "..", "sparsefusion_amd", "csrc")
# SF_EMU_DEFINES="A=1,B": extra -D flags for an A/B harness build (its own .so)
_DEFS = [d for d in os.environ.get("SF_EMU_DEFINES", "").split(",") if d]
_SO = os.path.join(_HERE, "_build", "libfused_emu" + "".join("_" + d.replace("=", "") for d in _DEFS) + ".so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
_handle = None


def lib():
    global _handle
    if _handle is None:
        srcs = [os.path.join(_HERE, f) for f in ("fused_emu.cpp", "hip_emu.h")] + \
               [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")] + \
               [os.path.join(_HERE, "..", "..", "include", "sparsefusion_hip.h")]
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.check_call([CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + _HERE, "-Wall", "-Wno-unused-function"] +
                                  ["-D" + d for d in _DEFS] + [os.path.join(_HERE, "fused_emu.cpp"), "-o", _SO, "-lpthread"])
        _handle = C.CDLL(_SO)
        _handle.emu_plan_run.restype = C.c_int
        _handle.emu_plan_run.argtypes = [C.POINTER(_lib.SfOp), C.c_uint32, C.c_char_p, C.c_int]
    return _handle


def available():
    return os.path.exists(CLANG)


def run(ops):
    arr = (_lib.SfOp * len(ops))(*ops)
    err = C.create_string_buffer(512)
    rc = lib().emu_plan_run(arr, len(ops), err, 512)
    if rc:
        raise RuntimeError(err.value.decode())


def mkop(type_, flags=0, p=(), i=(), f=()):
    o = _lib.SfOp()
    o.type, o.flags = type_, flags
    for k, v in enumerate(p):
        o.p[k] = (v.data_ptr() if isinstance(v, torch.Tensor) else v) if v is not None else None
    for k, v in enumerate(i):
        o.i[k] = int(v)
    for k, v in enumerate(f):
        o.f[k] = float(v)
    return o


def pack_conv_weights(w):
    """[Cout, Cin, kh, kw] fp32 -> MFMA fragment order bf16 (as int16 storage), the layout of sf_conv_pack_weights:
    [n_frag][tap * Cin/32 + cc][lane][8] with element (lane, j) = W[n_frag*16 + (lane & 15)][tap][cc*32 + 8*(lane >> 4) + j]."""
    co, ci, kh, kw = w.shape
    assert ci % 32 == 0
    nfr, cch, taps = (co + 15) // 16, ci // 32, kh * kw
    wp = torch.zeros(nfr * 16, ci, taps)
    wp[:co] = w.reshape(co, ci, taps)
    # [nf, n16, cc, kb, j, tap] -> [nf, tap, cc, kb, n16, j]
    x = wp.reshape(nfr, 16, cch, 4, 8, taps).permute(0, 5, 2, 3, 1, 4).contiguous()
    return x.to(torch.bfloat16).reshape(-1)
