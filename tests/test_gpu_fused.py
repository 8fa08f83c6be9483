"""GPU parity of the fused UNet ops through the C ABI (sf_plan_run): the cases of tests/fused_cases.py -- the small ones
the CPU-thread emulator also runs, and the UNet's own layer shapes -- against a torch fp32 reference on bf16-rounded
operands (tolerance 4e-3 relative L2: same operand rounding, different summation order)."""
import pytest

import fused_cases as fc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(fc.CONV_CASES) + sorted(fc.CONV_CASES_FULL))
def test_fused_conv_on_gpu(name):
    kw = fc.CONV_CASES.get(name) or fc.CONV_CASES_FULL[name]
    e = fc.run_conv_case("gpu", **kw)
    print(f"{name}: rel {e:.2e}")


def test_slots_kernel_and_gate_on_gpu():
    fc.run_slots_case("gpu")


@pytest.mark.parametrize("name", sorted({**fc.GCA_CASES, **fc.GCA_CASES_FULL}))
def test_gca_chain_on_gpu(name):
    fc.run_gca_case("gpu", **({**fc.GCA_CASES, **fc.GCA_CASES_FULL})[name])


@pytest.mark.parametrize("name", sorted(fc.ATTN_CASES) + ["unet_self_1024", "unet_cross_1024", "unet_b8_self_1024_wn2"])
def test_attention_prologue_on_gpu(name):
    """k_conv_fused<.., FNORM_ATTN>: the 16-token attention core in the prologue of its output projection (r04; was k_attn16 + conv)."""
    kw = fc.ATTN_CASES.get(name) or (dict(B=8, cross=False, Cout=1024, seed=69, WN=2) if name.endswith("wn2") else
                                     dict(B=1, cross=name == "unet_cross_1024", Cout=1024, seed=64))
    e = fc.run_attn_case("gpu", **kw)
    print(f"{name}: rel {e:.2e}")
