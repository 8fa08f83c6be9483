"""Build recipe for libsparsefusion_hip.so (gfx950) and the CPU oracle.

`python -m sparsefusion_amd.build` compiles every HIP source under csrc/ with
hipcc --offload-arch=gfx950 into ONE shared library that lives in-tree (so it
travels to the GPU box with the repo snapshot).  hipcc cross-compiles without
a GPU.  Objects are cached by source mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libsparsefusion_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, hdr_m, verbose):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_m):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj, True


def build(verbose=True, force=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_m = _deps_mtime()
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_m, verbose), _sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_variant(tag, defines, verbose=True, timing=False, sources=("unet_fused.hip",)):
    """A/B builds for GPU-side tuning: libsparsefusion_hip_<tag>.so = the product library with `sources` (default: unet_fused.hip)
    recompiled under extra -D flags (select it with SF_HIP_LIB=<path>); with timing=True the instrumented
    libsf_fused_timing_<tag>.so.  The software-dependent-launch experiment of DESIGN.md section 8 is
    build_variant("pdl", ["SF_PDL=1"], sources=("unet_fused.hip", "unet_ops.hip"))."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    dflags = ["-D" + d for d in defines]
    if timing:
        os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)
        out = os.path.join(ROOT, "tools", "_build", f"libsf_fused_timing_{tag}.so")
        srcs = [os.path.join(CSRC, f) for f in ("unet_fused.hip", "core.hip")]
        cmd = [HIPCC] + FLAGS + dflags + ["-DSF_FCONV_TIMING", "-shared", "-o", out] + srcs
    else:
        build(verbose=False)
        out = os.path.join(HERE, f"libsparsefusion_hip_{tag}.so")
        vobjs = []
        for src in sources:
            obj = os.path.join(OBJ_DIR, f"{src[:-4]}_{tag}.o")
            subprocess.check_call([HIPCC] + FLAGS + dflags + ["-c", os.path.join(CSRC, src), "-o", obj])
            vobjs.append(obj)
        objs = [os.path.join(OBJ_DIR, s[:-4] + ".o") for s in _sources() if s not in sources] + vobjs
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


LIB_F16 = os.path.join(HERE, "libsparsefusion_hip_f16.so")


def build_f16(verbose=True, force=False):
    """libsparsefusion_hip_f16.so: the whole library with IEEE-half MFMA operands instead of bf16 (-DSF_OPERAND_F16=1,
    csrc/sf_operand.h), selected per process with SF_OPERAND=f16 (BASELINE configs[4]: "fp16 UNet").  Objects are cached by
    source mtime like the default build's."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_m = _deps_mtime()

    def comp(src):
        obj = os.path.join(OBJ_DIR, src[:-4] + "_f16.o")
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_m):
            return obj, False
        cmd = [HIPCC] + FLAGS + ["-DSF_OPERAND_F16=1", "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(comp, _sources()))
    if any(c for _, c in res) or not os.path.exists(LIB_F16):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_F16] + [o for o, _ in res])
    return LIB_F16


def build_timing(verbose=True):
    """libsf_fused_timing.so: unet_fused.hip with in-kernel phase timestamps (-DSF_FCONV_TIMING), a measurement aid for
    tools/fconv_phases.py -- never loaded by the package."""
    os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)
    out = os.path.join(ROOT, "tools", "_build", "libsf_fused_timing.so")      # a measurement aid: not next to the product library
    srcs = [os.path.join(CSRC, f) for f in ("unet_fused.hip", "core.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    cmd = [HIPCC] + FLAGS + ["-DSF_FCONV_TIMING", "-shared", "-o", out] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--f16" in sys.argv:
        print(build_f16(force="--force" in sys.argv))
    if "--timing" in sys.argv:
        print(build_timing())
    print(LIB)
