"""Writes tests/golden/ngp_native.pt: outputs of the REFERENCE's own gridencoder.cu / raymarching.cu, compiled for the host
by oracle/build_ref.py, on the seeded cases of tests/native_cases.py.  Dev container only (needs /root/reference).

    python tests/golden/make_golden_native.py

Contents (inputs are regenerated from seeds, so the file holds reference OUTPUTS only):
  unfused/...   sha256 digests of every output of the -ffp-contract=off build: grid fwd / bwd for gridtype x D x C x
                align_corners (48 configs) and every raymarching entry point (3 march configs).  oracle/ngp_ref.c built
                -DORACLE_NO_FMA must reproduce every digest (bit equality).
  fused/...     the -ffp-contract=fast build (the nvcc -fmad=true analogue): digests of the D = 3 grid configs and of the
                marching / near-far / morton / packbits outputs (bit equality with the default oracle build), tensors of the
                sums-of-products outputs (tolerance 1e-6: g++ and the oracle's explicit fmaf do not contract the same sums).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import native_cases as nc  # noqa: E402
from oracle import ref_native  # noqa: E402


def main():
    assert ref_native.available(), "needs /root/reference (oracle/build_ref.py)"
    gold = {"unfused": {}, "fused": {}, "fused_tensors": {}}
    with ref_native.unfused():
        for cfg in nc.GRID_CONFIGS:
            r = nc.run_grid(ref_native, nc.grid_case(*cfg))
            gold["unfused"][("grid",) + cfg] = {k: nc.digest(v) for k, v in r.items()}
        for cfg in nc.MARCH_CONFIGS:
            r = nc.run_raymarching(ref_native, *cfg)
            gold["unfused"][("rm",) + cfg] = {k: nc.digest(v) for k, v in r.items()}
    for cfg in nc.GRID_CONFIGS:
        if cfg[1] == 3:
            r = nc.run_grid(ref_native, nc.grid_case(*cfg))
            gold["fused"][("grid",) + cfg] = {k: nc.digest(v) for k, v in r.items()}
    for cfg in nc.MARCH_CONFIGS:
        r = nc.run_raymarching(ref_native, *cfg)
        skip = nc.FUSED_TOLERANT + nc.FUSED_DEPENDENT
        gold["fused"][("rm",) + cfg] = {k: nc.digest(v) for k, v in r.items() if k not in skip}
        gold["fused_tensors"][("rm",) + cfg] = {k: r[k].clone() for k in nc.FUSED_TOLERANT}
    out = os.path.join(HERE, "ngp_native.pt")
    torch.save(gold, out)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
