"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and the product package never touches the oracle (no CPU fallback)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            src = open(os.path.join(inc, f)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(sf_[a-zA-Z0-9_]+)\s*\(", src))
    return names


def test_library_built_and_exports_header_symbols():
    from sparsefusion_amd import build, _lib
    build.build(verbose=False)
    import ctypes
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 10
    missing = [n for n in sorted(declared) if not hasattr(handle, n)]
    assert not missing, f"header declares symbols the .so lacks: {missing}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib().sf_abi_version() >= 1
    # the IEEE-half operand build (BASELINE configs[4] "fp16 UNet") is the same ABI: every symbol, and it says what it multiplies
    f16 = ctypes.CDLL(build.build_f16(verbose=False))
    assert not [n for n in sorted(declared) if not hasattr(f16, n)]
    assert f16.sf_operand_is_f16() == 1 and handle.sf_operand_is_f16() == 0
    assert _lib.lib("f16").sf_operand_is_f16() == 1 and _lib.lib("bf16").sf_operand_is_f16() == 0


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, "sparsefusion_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt \
                        or re.search(r"#include\s+[\"<].*oracle", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_no_cpu_fallback_for_cpu_tensors():
    import torch
    from sparsefusion_amd import raymarching
    with pytest.raises(RuntimeError):
        raymarching.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]), 0.1)


def test_missing_library_fails_loudly():
    code = ("import sparsefusion_amd._lib as L, os; L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'nope.so');\n"
            "try:\n    L.lib()\nexcept RuntimeError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_grid_offsets_match_reference_layout():
    from sparsefusion_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                      desired_resolution=8192, gridtype='tiled')
    assert tuple(enc.embeddings.shape) == (929336, 2)
    assert enc.offsets.dtype.is_floating_point is False and enc.offsets.numel() == 17
    assert list(enc.host_offsets[:4]) == [0, 4920, 22496, 77368]
    assert set(enc.state_dict().keys()) == {"embeddings", "offsets"}


def test_ngp_state_dict_keys():
    from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
    net = NeRFNetwork(get_default_torch_ngp_opt())
    want = {"aabb_train", "aabb_infer", "encoder.embeddings", "encoder.offsets"} | {
        f"sigma_net.net.{i}.{w}" for i in range(3) for w in ("weight", "bias")}
    assert set(net.state_dict().keys()) == want
    groups = net.get_params(5e-4)
    assert groups[0]["lr"] == pytest.approx(5e-3) and groups[1]["lr"] == pytest.approx(5e-4)


def test_render_backward_workspace_is_bounded_for_any_ray_count():
    """r05: the bins of the table-gradient scatter are sized by the largest ray chunk of the backward, and a ray count
    without an equal split (200^2 = 40 000, 65 535) used to make ONE chunk of all rays (7.9 GB / 13 GB of bins).  Chunks now hold at
    most 8192 rays for any N (host-side arithmetic of the C ABI: no GPU needed)."""
    from sparsefusion_amd import _lib
    lib = _lib.lib()
    T = 64
    per_sample = 72 * T * 4                                            # composite / field gradients: 72 N T floats
    bins_8192 = 8192 * 2 * T * 4 * 24 * 16                             # 96 sixteen-byte entries per sample of one 8192-ray chunk
    for N in (16384, 40000, 65535, 10000, 8192 * 3 + 1, 51712):        # 51 712 = 2^9 x 101: its only equal split is 2 x 25 856 rays (r06)
        wb = lib.sf_ngp_render_workspace_bytes(N, T)
        assert wb <= N * per_sample + bins_8192 + (1 << 24), (N, wb)            # + cursors and per-bucket slack (8.4 MB)
    assert lib.sf_ngp_render_workspace_bytes(256, T) < 256 * per_sample + 256 * 2 * T * 4 * 24 * 16 + (1 << 24)
