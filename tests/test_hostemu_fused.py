"""Kernel-logic tests of the fused UNet kernels (sparsefusion_amd/csrc/fused_kernels.h) on CPU threads: the product's
kernel source is compiled by the host clang with one fiber per lane (tests/hostemu/hip_emu.h) and compared with a
plain torch fp32 reference of the same op on bf16-rounded operands (cases: tests/fused_cases.py).  The GPU launch path
runs the same cases in tests/test_gpu_fused.py."""
import os

import pytest
import torch

import fused_cases as fc
from hostemu import fused

pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


# 15-45 s of thread emulation each: by default only on the GPU (tests/test_gpu_fused.py runs every case); SF_SLOW_TESTS=1 runs them here too
HEAVY_ON_CPU = {"pipe_pool_16x16_tr2", "gn_slots_xcd_map_16x16", "pipe_gn_slots_16x16_tr2_wn2", "pipe_pool_16x16_wn2", "pipe_pair_gn_slots_16x16_tr2"}


@pytest.mark.parametrize("name", sorted(fc.CONV_CASES))
def test_fused_conv_on_cpu_threads(name):
    if name in HEAVY_ON_CPU and not os.environ.get("SF_SLOW_TESTS"):
        pytest.skip("heavy emulation case: SF_SLOW_TESTS=1 (covered on the GPU by tests/test_gpu_fused.py)")
    fc.run_conv_case("emu", **fc.CONV_CASES[name])


def test_slots_kernel_and_gate():
    fc.run_slots_case("emu")


@pytest.mark.parametrize("dim", [64, 128])
def test_whole_fused_plan_against_oracle(dim):
    """The complete fused launch plan of a dim-64 (mixed fused / first-round ops) and a dim-128 (every block fused) UNet,
    interpreted on the CPU (hostemu/plan_interp.py), against the fp32 oracle: planner wiring, workspaces, lazy tensors.  (An hour on
    the thread-per-lane emulator of r01-r03 and therefore opt-in; 16 s / 60 s on the fiber scheduler of r04: part of the regular suite.)"""
    from oracle import unet_ref
    from sparsefusion_amd.unet import Unet, _Plan, unet_param_spec
    from hostemu import plan_interp
    cfg = dict(dim=dim, dim_mults=(1, 2, 4, 4), num_resnet_blocks=2, layer_attns=(False, False, False, True),
               cond_images_channels=28, channels=4)
    net = Unet(**cfg, layer_cross_attns=(False,) * 4, attn_pool_text=False)
    sd = unet_ref.init_state(unet_param_spec(dim, (1, 2, 4, 4), net.nres, net.attns, 28, 4, net.cond_dim), seed=0)
    net.load_state_dict(sd, strict=True)
    cpu = torch.device("cpu")
    s = _Plan(net, 1, cpu).build()
    plan = _Plan(net, 1, cpu, (s.zero.off, s.misc.off + s.ws_bytes + s.ws2_bytes + 512, s.ws_bytes, s.ws2_bytes)).build()
    g = torch.Generator().manual_seed(5)
    x, cond = torch.randn(1, 4, 32, 32, generator=g), torch.randn(1, 28, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.tensor([0.37]))
    plan.misc.buf.view(torch.float32)[:] = float("nan")               # a read of an unwritten activation would show
    plan.x_view.copy_(x.reshape(1, -1)); plan.t_view.copy_(ls.reshape(1, 1)); plan.cond_view.copy_(cond.reshape(1, -1))
    plan_interp.run_plan(plan.ops)
    y = plan.out_view.clone().view(1, 4, 32, 32)
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond)
    assert torch.isfinite(y).all() and fc.rel(y, y_ref) < 2e-2


@pytest.mark.parametrize("name", sorted(fc.GCA_CASES))
def test_gca_chain_on_cpu_threads(name):
    fc.run_gca_case("emu", **(fc.GCA_CASES)[name])


@pytest.mark.parametrize("name", sorted(fc.ATTN_CASES))
def test_attention_prologue_on_cpu_threads(name):
    fc.run_attn_case("emu", **fc.ATTN_CASES[name])


def test_pipelined_pairs_run_the_merged_res_conv_kernel():
    """conv1 || res_conv of the pipelined tiles with WM <= 2 run as ONE set of workgroups (k_conv_fused_pipe_rc, r04): the pair case
    above must actually have taken that kernel; the 64-pixel tile (WM = 4) keeps the two-kernel launch."""
    n0 = fused.lib().emu_rc_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES["pipe_pair_gn_slots_concat_8x8"])
    n1 = fused.lib().emu_rc_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES["pipe_pair_gn_slots_wide_rows_wm4"])
    assert n1 == n0 + 1 and fused.lib().emu_rc_launches() == n1


def test_the_4x4_geometry_runs_on_k_conv4_gn():
    """r05: an op of the geometry csrc/fused_conv4.h is written for must take that kernel (fused_host.h::conv4_cs4, the dispatch
    unet_fused.hip::run_fconv shares with this harness); op flag 128 and every other 4x4 shape keep k_conv_fused."""
    n0 = fused.lib().emu_conv4_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES["conv4_gn_lazy_splitk_1024"])
    n1 = fused.lib().emu_conv4_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES["conv4_geometry_on_the_general_kernel"])
    fc.run_conv_case("emu", **fc.CONV_CASES["gn_self_sliced_lazy_splitk_4x4"])
    assert n1 == n0 + 1 and fused.lib().emu_conv4_launches() == n1
    m0 = fused.lib().emu_conv4_mb_launches()                                       # B >= 2: several images per workgroup (k_conv4_gn_mb)
    fc.run_conv_case("emu", **fc.CONV_CASES["conv4_mb_lazy_splitk_1024_b4"])
    fc.run_conv_case("emu", **fc.CONV_CASES["conv4_gn_plain_1024_b2"])
    m1 = fused.lib().emu_conv4_mb_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES["conv4_one_image_per_workgroup_b4"])    # op field i[19] bit 0 keeps k_conv4_gn
    assert m1 == m0 + 2 and fused.lib().emu_conv4_mb_launches() == m1 and fused.lib().emu_conv4_launches() == n1 + 1
    n1 += 1
    fc.run_conv_case("emu", **fc.CONV_CASES["conv4_no_norm_plain_1024"])        # r06: the un-normalised form of the geometry (k_conv4_gn<64, 0, false>)
    assert fused.lib().emu_conv4_launches() == n1 + 1
    n1 += 1
    fc.run_conv_case("emu", **fc.CONV_CASES["lin4_ln_ff1_1024_gelu"])           # k_lin4_ln counts on the same counter
    n2 = fused.lib().emu_conv4_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES["lin4_shape_on_the_general_kernel"])
    fc.run_conv_case("emu", **fc.CONV_CASES["layernorm_linear"])
    assert n2 == n1 + 1 and fused.lib().emu_conv4_launches() == n2
    fc.run_attn_case("emu", **fc.ATTN_CASES["self_plain"])                       # r06: k_lin4_attn counts on the same counter
    n3 = fused.lib().emu_conv4_launches()
    fc.run_attn_case("emu", **fc.ATTN_CASES["self_context_on_the_general_kernel"])
    assert n3 == n2 + 1 and fused.lib().emu_conv4_launches() == n3


CONV3S_ON_CPU = sorted(n for n in fc.CONV_CASES_FULL if n.startswith("conv3s_"))


@pytest.mark.parametrize("name", CONV3S_ON_CPU)
def test_conv3s_on_cpu_threads(name):
    """r06: k_conv3s (csrc/fused_conv3s.h) exists for the UNet's own layer shapes only, so its emulation cases are full-size (1-3 s each on
    the fiber scheduler): every instantiated tile family -- strips and 2-D tiles, WN = 1 / 2 / 4, with residual, scale-shift, epilogue pooling
    -- and the conv1 + res_conv pairs on two sources (k_conv3s_rc) against the torch reference; the op must have taken that kernel (and the
    general one under op field i[19] bit 1)."""
    n0 = fused.lib().emu_conv3s_launches()
    fc.run_conv_case("emu", **fc.CONV_CASES_FULL[name])
    took = fused.lib().emu_conv3s_launches() - n0
    assert took == (0 if fc.CONV_CASES_FULL[name].get("keep_pipe") else 1)

