"""Builds oracle/_ref/libref_native.so: the REFERENCE's own CUDA sources for the natives of the hot path, compiled for the
host CPU.  TEST INFRASTRUCTURE ONLY (it pins oracle/ngp_ref.c; nothing under sparsefusion_amd/ may touch it).

    python oracle/build_ref.py          (dev container only: needs /root/reference; the GPU box uses the prebuilt .so)

What it does: reads /root/reference/external/gridencoder/src/gridencoder.cu and /root/reference/raymarching/src/raymarching.cu
WHERE THEY LIE, rewrites exactly one construct that is not C++ -- `kernel<<<grid, block>>>(args);` becomes
`shim::launch(grid, block, [&]{ kernel(args); });` -- and pipes the text to g++ on stdin (no copy of a reference source is
written anywhere) with oracle/cuda_shim/ in front of the include path: `__global__` / `__device__` vanish, `threadIdx` /
`blockIdx` are thread-local globals, a launch is a serial loop over the grid, `atomicAdd` is a read-modify-write, ATen is
the pointer-and-dtype struct the host wrappers need, AT_DISPATCH instantiates float only.  The reference's own build
system (setup.py / torch cpp_extension / nvcc) is not run.

Deviations from nvcc that the pinned tests must (and do) allow for:
  * `__expf` -> expf (nvcc: ex2.approx, 2 ulp): the composite kernels' weights carry a few-ulp tolerance;
  * FMA contraction: nvcc -fmad=true contracts every a*b+c it sees; g++ -ffp-contract=fast -mfma contracts the same
    expressions in straight-line code (both are "contract wherever syntactically possible" in a single expression);
  * exp2f / powf / frexpf are glibc's (correctly rounded) instead of CUDA's libdevice (<= 2 ulp).
Integer outputs (cell indices, hash rows, morton codes, bitfields, ray / point slots, step counts) are unaffected by all three
except through sample positions that sit within an ulp of a cell wall.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SF_REFERENCE_ROOT", "/root/reference")
SOURCES = ("external/gridencoder/src/gridencoder.cu", "raymarching/src/raymarching.cu")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libref_native.so")                 # -ffp-contract=fast: the nvcc -fmad=true analogue
OUT_UNFUSED = os.path.join(OUT_DIR, "libref_native_nofma.so")    # -ffp-contract=off: no FMA anywhere
CXX = os.environ.get("CXX", "g++")
CXXFLAGS = ["-O2", "-fPIC", "-std=c++17", "-mfma", "-mavx2", "-w",
            "-I" + os.path.join(HERE, "cuda_shim")]

_LAUNCH = re.compile(r"([A-Za-z_]\w*(?:<[^<>;]*>)?)\s*<<<(.*?)>>>\s*\((.*?)\)\s*;", re.S)


def _split_top(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [o.strip() for o in out]


def rewrite_launches(text):
    def sub(m):
        kern, cfg, args = m.group(1), _split_top(m.group(2)), m.group(3)
        assert len(cfg) == 2, cfg                       # <<<grid, block>>> only: no dynamic LDS, no stream
        return f"shim::launch({cfg[0]}, {cfg[1]}, [&]{{ {kern}({args}); }});"
    out, n = _LAUNCH.subn(sub, text)
    assert "<<<" not in out, "unhandled launch syntax"
    return out, n


def available():
    return all(os.path.exists(os.path.join(REF, s)) for s in SOURCES)


def build(force=False, verbose=False, unfused=False):
    """Returns the path of the library; builds it when the reference is present and the library is missing or older than
    its inputs; returns None when neither the reference nor a prebuilt library exists."""
    OUT = OUT_UNFUSED if unfused else globals()["OUT"]
    contract = ["-ffp-contract=off"] if unfused else ["-ffp-contract=fast"]
    if not available():
        return OUT if os.path.exists(OUT) else None
    deps = [os.path.join(REF, s) for s in SOURCES] + [os.path.join(HERE, "ref_exports.cpp"), os.path.abspath(__file__)]
    for root, _, files in os.walk(os.path.join(HERE, "cuda_shim")):
        deps += [os.path.join(root, f) for f in files]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    for s in SOURCES:
        text, n = rewrite_launches(open(os.path.join(REF, s)).read())
        obj = os.path.join(OUT_DIR, os.path.basename(s)[:-3] + ("_nofma.o" if unfused else ".o"))
        if verbose:
            print(f"{s}: {n} launches rewritten -> {obj}")
        subprocess.run([CXX] + CXXFLAGS + contract + ["-x", "c++", "-c", "-", "-o", obj], input=text.encode(), check=True)
        objs.append(obj)
    incs = ["-I" + os.path.join(REF, os.path.dirname(s_)) for s_ in SOURCES]          # the reference's own headers, where they lie
    subprocess.check_call([CXX] + CXXFLAGS + contract + incs + ["-shared", os.path.join(HERE, "ref_exports.cpp")] + objs + ["-o", OUT, "-lm"])
    for o in objs:
        os.remove(o)
    return OUT


if __name__ == "__main__":
    for uf in (False, True):
        p = build(force="--force" in sys.argv, verbose=True, unfused=uf)
        print(p if p else "reference sources absent and no prebuilt library")
