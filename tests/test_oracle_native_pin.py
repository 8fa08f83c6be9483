"""Pins oracle/ngp_ref.c (the C restatement every GPU parity test of the NGP natives compares against) to the REFERENCE's own
CUDA sources: external/gridencoder/src/gridencoder.cu:30-479 and raymarching/src/raymarching.cu:56-913 compiled for the host
(oracle/build_ref.py -> oracle/_ref/).  Two layers:
  * golden: tests/golden/ngp_native.pt (made from the reference by tests/golden/make_golden_native.py) -- runs everywhere;
  * live:   the same comparison tensor by tensor against oracle/_ref/libref_native*.so when it is present (dev container,
            or the GPU box, where the prebuilt library travels with the snapshot), and a check that the golden file is what
            that library produces today.
Bit equality for the contraction-free builds on every configuration; bit equality for the default (FMA) builds wherever
g++ and nvcc have no freedom (D = 3 grid, marching, indices); 1e-6 on the compositing sums."""
import contextlib

import pytest
import torch

import native_cases as nc
from oracle import ngp_native, ref_native


@pytest.fixture(scope="module")
def gold(golden_dir):
    return torch.load(f"{golden_dir}/ngp_native.pt")


@pytest.mark.parametrize("cfg", nc.GRID_CONFIGS, ids=lambda c: "t%d-D%d-C%d-ac%d" % c)
def test_grid_unfused_oracle_reproduces_reference_digests(gold, cfg):
    with ngp_native.unfused():
        got = nc.run_grid(ngp_native, nc.grid_case(*cfg))
    want = gold["unfused"][("grid",) + cfg]
    assert {k: nc.digest(v) for k, v in got.items()} == want


@pytest.mark.parametrize("cfg", [c for c in nc.GRID_CONFIGS if c[1] == 3], ids=lambda c: "t%d-D%d-C%d-ac%d" % c)
def test_grid_default_oracle_reproduces_contracted_reference_digests_3d(gold, cfg):
    got = nc.run_grid(ngp_native, nc.grid_case(*cfg))
    assert {k: nc.digest(v) for k, v in got.items()} == gold["fused"][("grid",) + cfg]


@pytest.mark.parametrize("cfg", nc.MARCH_CONFIGS, ids=lambda c: "dtg%g-perturb%d" % c)
def test_raymarching_unfused_oracle_reproduces_reference_digests(gold, cfg):
    with ngp_native.unfused():
        got = nc.run_raymarching(ngp_native, *cfg)
    want = gold["unfused"][("rm",) + cfg]
    bad = [k for k in want if nc.digest(got[k]) != want[k]]
    assert not bad and set(got) == set(want), bad


@pytest.mark.parametrize("cfg", nc.MARCH_CONFIGS, ids=lambda c: "dtg%g-perturb%d" % c)
def test_raymarching_default_oracle_vs_contracted_reference(gold, cfg):
    got = nc.run_raymarching(ngp_native, *cfg)
    want = gold["fused"][("rm",) + cfg]
    bad = [k for k in want if nc.digest(got[k]) != want[k]]
    assert not bad, bad                                              # sample positions, deltas, slots, counters, bitfields: exact
    for k, ref in gold["fused_tensors"][("rm",) + cfg].items():
        a, b = got[k].double(), ref.double()
        assert torch.equal(torch.isnan(a), torch.isnan(b)), k
        m = ~torch.isnan(a)
        assert float((a[m] - b[m]).abs().max()) <= 1e-6 * max(1.0, float(b[m].abs().max())), k


def test_reference_host_wrappers_reject_what_they_reject():
    """The reference's own TORCH_CHECK / runtime_error paths run in the host build (gridencoder.cu:347-353,:361-369)."""
    if not ref_native.available():
        pytest.skip("oracle/_ref absent (needs /root/reference to build)")
    c = nc.grid_case(0, 3, 2, False)
    out = torch.zeros(c["L"], c["B"], 3)
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        ref_native.grid_encode_forward(c["x"], c["emb"], c["offsets"], out, c["B"], 3, 3, c["L"], c["S"], c["H"], None, 0, False)
    with pytest.raises(RuntimeError, match="D must be"):
        ref_native.grid_encode_forward(c["x"], c["emb"], c["offsets"], out, c["B"], 6, 2, c["L"], c["S"], c["H"], None, 0, False)


@pytest.mark.parametrize("unfused", [True, False], ids=["contract-off", "contract-fast"])
def test_live_reference_library_matches_oracle_and_golden(gold, unfused):
    if not ref_native.available():
        pytest.skip("oracle/_ref absent (needs /root/reference to build)")
    rctx = ref_native.unfused if unfused else contextlib.nullcontext
    octx = ngp_native.unfused if unfused else contextlib.nullcontext
    key = "unfused" if unfused else "fused"
    for cfg in nc.GRID_CONFIGS:
        if not unfused and cfg[1] != 3:
            continue
        case = nc.grid_case(*cfg)
        with rctx():
            r = nc.run_grid(ref_native, case)
        with octx():
            o = nc.run_grid(ngp_native, case)
        for k in r:
            assert torch.equal(r[k], o[k]), (cfg, k, float((r[k] - o[k]).abs().max()))
        assert {k: nc.digest(v) for k, v in r.items()} == gold[key][("grid",) + cfg], ("stale golden", cfg)
    for cfg in nc.MARCH_CONFIGS:
        with rctx():
            r = nc.run_raymarching(ref_native, *cfg)
        with octx():
            o = nc.run_raymarching(ngp_native, *cfg)
        for k, want in gold[key][("rm",) + cfg].items():
            assert torch.equal(torch.nan_to_num(r[k].double(), nan=1e30), torch.nan_to_num(o[k].double(), nan=1e30)), (cfg, k)
            assert nc.digest(r[k]) == want, ("stale golden", cfg, k)
