"""GPU parity of the full UNet forward and the PLMS sampler: HIP plan vs the fp32 oracle and the golden
outputs of the real reference.  Compute dtype is bf16 operands / fp32 accumulate (the reference loop is fp32),
so the tolerance is a stated bf16 one: relative L2 error < 2e-2 and cosine > 0.9995 per eval."""
import pytest
import torch

from oracle import unet_ref
from unet_common import CONFIGS, GOLD, cosine, inputs, rel_err, spec, state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_REL, TOL_COS = 2e-2, 0.9995


def _unet(name, sd=None):
    from sparsefusion_amd.unet import Unet
    net = Unet(**CONFIGS[name], layer_cross_attns=(False,) * 4, attn_pool_text=False)
    missing = net.load_state_dict(sd if sd is not None else state(name), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.to(DEV)


@pytest.mark.parametrize("name", ["small", "canonical"])
def test_unet_forward_matches_reference_golden(name):
    g = torch.load(f"{GOLD}/unet_forward.pt")[name]
    net = _unet(name)
    x, ls, cond = inputs(CONFIGS[name], g["B"], g["input_seed"])
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_scale=1.).cpu()
    assert y.shape == g["y"].shape and torch.isfinite(y).all()
    r, c = rel_err(y, g["y"]), cosine(y, g["y"])
    print(f"unet[{name}] rel L2 err {r:.3e}  cosine {c:.6f}  max abs {float((y - g['y']).abs().max()):.3e}")
    assert r < TOL_REL and c > TOL_COS
    # batch-consistent: a sample evaluated alone takes other tile / split-K shapes (different fp32 summation order
    # in front of every bf16 rounding), so it agrees to the same bf16 tolerance, not bit-wise
    y0 = net.forward_with_cond_scale(x[:1].to(DEV), ls[:1].to(DEV), cond_images=cond[:1].to(DEV)).cpu()
    assert rel_err(y0, y[:1]) < TOL_REL and rel_err(y0, g["y"][:1]) < TOL_REL


def test_unet_medium_config_runs_the_canonical_kernels_and_matches_reference():
    """dim 128: every block of this configuration plans onto the fused kernels the 400 M-parameter model runs (slot statistics,
    fixed-order 4x4 sums, pooled fragments, attention prologue: no atomics anywhere, unlike dim 64) -- a model-level parity case
    that runs the SAME reductions as the benchmarked plan: forward golden, 51-eval PLMS trajectory golden (B = 2), bitwise
    reproducibility."""
    from sparsefusion_amd.unet import OP_GN_ACT, OP_GCA_POOL, OP_ATTN
    from sparsefusion_amd.vldm import DDPM
    from sparsefusion_amd.plms import PLMSSampler
    name = "medium"
    G = torch.load(f"{GOLD}/unet_medium.pt")
    g = G["forward"]
    net = _unet(name)
    x, ls, cond = inputs(CONFIGS[name], g["B"], g["input_seed"])
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_scale=1.).cpu()
    plan = net._plan(g["B"], torch.device(DEV))
    assert plan.zero.off == 0 and not any(o.type in (OP_GN_ACT, OP_GCA_POOL, OP_ATTN) for o in plan.ops)     # fully fused, nothing accumulates with atomics
    r, c = rel_err(y, g["y"]), cosine(y, g["y"])
    print(f"unet[medium] rel L2 err {r:.3e}  cosine {c:.6f}")
    assert r < TOL_REL and c > TOL_COS
    y2 = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_scale=1.).cpu()
    assert torch.equal(y, y2)
    rr = G["plms"]
    vldm = DDPM(channels=4, unets=(net,), conditional_encoder=None, conditional_embed_dim=None, image_sizes=(32,),
                timesteps=500, cond_drop_prob=0.1, pred_objectives='noise', conditional=False, auto_normalize_img=False,
                clip_output=True, dynamic_thresholding=False, dynamic_thresholding_percentile=.68, clip_value=10).to(DEV)
    gg = torch.Generator().manual_seed(rr["input_seed"])
    lat = 0.5 * torch.randn(2, 4, 32, 32, generator=gg)
    cond = torch.randn(2, CONFIGS[name]["cond_images_channels"], 32, 32, generator=gg)
    torch.manual_seed(rr["noise_seed"])
    noises = [torch.randn(2, 4, 32, 32).to(DEV) for _ in range(unet_ref.plms_noise_count(rr["max_thres"]))]
    imgs = []
    for _ in range(2):
        img, xn, nz, acp = PLMSSampler(vldm, 50).sample(lat.to(DEV), cond_images=cond.to(DEV), use_tqdm=False, return_noise=True,
                                                        max_thres=rr["max_thres"], noises=noises)
        imgs.append(img.cpu())
    e, cc = rel_err(imgs[0], rr["img"]), cosine(imgs[0], rr["img"])
    print(f"plms medium (51 evals, B=2) rel {e:.3e} cos {cc:.6f}")
    assert torch.equal(nz.cpu(), rr["noise"]) and e < 3e-2 and cc > 0.9995
    assert torch.equal(imgs[0], imgs[1])                          # a seeded trajectory is reproducible bit for bit


def test_unet_fp16_operands_match_reference_golden():
    """BASELINE configs[4] "fp16 UNet": the same plan on the IEEE-half operand build of the library (csrc/sf_operand.h,
    libsparsefusion_hip_f16.so: v_mfma_f32_16x16x32_f16, fp32 accumulate) against the fp32 reference golden -- selected per module
    with set_operand("f16") or, as a reference user would, with .half()."""
    name = "canonical"
    g = torch.load(f"{GOLD}/unet_forward.pt")[name]
    net = _unet(name).set_operand("f16")
    assert net.clib.sf_operand_is_f16() == 1
    x, ls, cond = inputs(CONFIGS[name], g["B"], g["input_seed"])
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_scale=1.).cpu()
    r, c = rel_err(y, g["y"]), cosine(y, g["y"])
    print(f"unet[{name}] fp16 operands: rel L2 err {r:.3e}  cosine {c:.6f}")
    assert torch.isfinite(y).all() and r < TOL_REL and c > TOL_COS
    net.set_operand("bf16")                                          # same module on the bf16 build: new plans, other roundings
    yb = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_scale=1.).cpu()
    assert rel_err(yb, g["y"]) < TOL_REL and not torch.equal(yb, y)
    # .half(): parameters in fp16 -> fp16 operands, half in, half out; the sampler fast path runs on the same build
    name = "small"
    gs = torch.load(f"{GOLD}/unet_forward.pt")[name]
    small = _unet(name).half()
    assert small.operand == "f16"
    x, ls, cond = inputs(CONFIGS[name], gs["B"], gs["input_seed"])
    yh = small.forward_with_cond_scale(x.to(DEV).half(), ls.to(DEV), cond_images=cond.to(DEV).half(), cond_scale=1.)
    assert yh.dtype == torch.float16 and rel_err(yh.float().cpu(), gs["y"]) < TOL_REL
    ctx = small.begin_sampling(cond.to(DEV), ls.to(DEV))
    z = small.eval_prepared(ctx, x.to(DEV), 0).clone().cpu()
    assert rel_err(z[:1], gs["y"][:1]) < TOL_REL


def test_unet_layerwise_against_oracle():
    """Intermediate activations (down/mid/up stages) of the small config vs the oracle: localises errors."""
    name = "small"
    sd = state(name)
    net = _unet(name, sd)
    x, ls, cond = inputs(CONFIGS[name], 1, 21)
    probe = {}
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond, probe=probe)
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
    assert rel_err(y, y_ref) < TOL_REL


@pytest.mark.parametrize("B", [1, 4])
def test_lazy_consumers_do_not_change_the_result(B):
    """Split-K partials / gated residuals materialised by their first consumer (GroupNorm statistics pass, gca
    logits pass) instead of their own launch.  Element values are the same sums in the same order, but the
    statistics pass slices its fp32 partial sums differently, so means differ in the last ulp and bf16 roundings
    downstream decorrelate: the two plans agree to the bf16 tolerance and BOTH meet the oracle tolerance.
    B = 4 is the views-per-GPU batch of BASELINE config 4."""
    name = "canonical"
    sd = state(name)
    net = _unet(name, sd)
    g = torch.Generator().manual_seed(33)
    x = torch.randn(B, 4, 32, 32, generator=g)
    cond = torch.randn(B, 256, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.tensor([0.8, 0.3, 0.55, 0.02][:B]))
    ys = {}
    for mode in (0, 3):
        net.lazy_consumers = mode
        net.invalidate()
        ys[mode] = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
        n_ops = len(net._plan(B, torch.device(DEV)).ops)
        print(f"B={B} lazy_consumers={mode}: {n_ops} ops")
    assert rel_err(ys[3], ys[0]) < TOL_REL
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond)
    for mode in (0, 3):
        r, c = rel_err(ys[mode], y_ref), cosine(ys[mode], y_ref)
        print(f"B={B} lazy_consumers={mode} vs oracle: rel L2 {r:.3e} cosine {c:.6f}")
        assert r < TOL_REL and c > TOL_COS


@pytest.mark.parametrize("switch", ["hybrid_rows", "no_big_tiles", "cost_model_tiles"])
def test_plan_switches_match_oracle(switch):
    """Every planner switch that changes which kernels an eval runs has its own parity case (the default suite runs B = 1, 2, 4 on
    the default plan only): `unfused_min_rows` (ResnetBlocks with >= that many pixel rows at 32x32 / 16x16 leave the GroupNorm-fused
    kernels for GroupNorm + k_conv3_halo: the default threshold 8192 is first reached at B = 8; forced here at B = 2),
    `big_tile_min_batch` (the 32- / 64-pixel tiles from B = 2 on, here switched off) and `tile_override` (measured implicit-GEMM tile
    picks, here the cost model's).  Each plan is compared with the fp32 oracle and with the default plan."""
    name = "canonical"
    sd = state(name)
    net = _unet(name, sd)
    B = 2
    g = torch.Generator().manual_seed(91)
    x, cond = torch.randn(B, 4, 32, 32, generator=g), torch.randn(B, 256, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.tensor([0.7, 0.15]))
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond)
    y_def = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
    n_def = [o.type for o in net._plan(B, torch.device(DEV)).ops]
    if switch == "hybrid_rows":
        net.unfused_min_rows = net.unfused_min_rows_16 = 512       # (r06: the 16x16 level has a threshold of its own)
    elif switch == "no_big_tiles":
        net.big_tile_min_batch = 999
    else:
        net.tile_override = {}
    net.drop_plans()
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
    ops = net._plan(B, torch.device(DEV)).ops
    if switch == "hybrid_rows":                    # the plan really changed: GroupNorm passes + LDS-tiled convs appeared
        from sparsefusion_amd.unet import OP_CONV, OP_GN_ACT
        assert sum(o.type == OP_GN_ACT for o in ops) >= 16 and sum(o.type == OP_CONV and o.i[14] >= 256 for o in ops) >= 12
        assert [o.type for o in ops] != n_def
    elif switch == "no_big_tiles":
        from sparsefusion_amd.unet import OP_FCONV
        assert max(o.i[15] for o in ops if o.type == OP_FCONV) <= 2          # WM: no 64-pixel tiles
    r, c = rel_err(y, y_ref), cosine(y, y_ref)
    print(f"{switch}: rel L2 vs oracle {r:.3e} cosine {c:.6f}; vs default plan {rel_err(y, y_def):.3e}")
    assert torch.isfinite(y).all() and r < TOL_REL and c > TOL_COS and rel_err(y, y_def) < TOL_REL


@pytest.mark.parametrize("plan", ["default", "r05"])
@pytest.mark.parametrize("B", [8, 16, 32])
def test_large_batch_default_plans_match_oracle(B, plan):
    """The plans the strong-scaling lines are timed on (`also_measured.config3_total32`: 32 views on one GPU; N = 2 / 4 ranks: 16 / 8 per
    GPU) against the fp32 oracle with the same bounds as B = 1 (imagen_pytorch.py:1470-1671).  These batch sizes are the first to reach
    the hybrid ResnetBlocks (GroupNorm pass + k_conv3_halo from 8192 pixel rows on), the WN = 2 attention-prologue projection and -- r06, the
    default -- the 4x4 level on the LDS-tiled kernels with split-K groups and the Upsample convs with the pixel shuffle on the LDS-tiled
    kernels; "r05" switches the last two off (k_conv4_gn_mb with 4 images per workgroup and its own split-K reduction launches; still what
    B <= 4 runs, and a fallback).  The test asserts the plan contains the kernels it names, so that a planner change cannot silently move the
    timed plan off the tested kernels."""
    from sparsefusion_amd import unet as U
    name = "canonical"
    sd = state(name)
    net = _unet(name, sd)
    if plan == "r05":
        net.lds_mid_min_rows = net.unfused_min_rows_4 = net.unfused_min_rows_8 = 0
    g = torch.Generator().manual_seed(600 + B)
    x, cond = torch.randn(B, 4, 32, 32, generator=g), torch.randn(B, 256, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.rand(B, generator=g) * 0.98 + 0.01)
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond)
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
    ops = net._plan(B, torch.device(DEV)).ops
    halo = [o for o in ops if o.type == U.OP_CONV and o.i[14] >= 256 and o.i[9] == 3 and o.i[1] >= 16]
    gn_pass = [o for o in ops if o.type == U.OP_GN_ACT]
    c4 = [o for o in ops if o.type == U.OP_FCONV and o.i[12] == U.FNORM_GN_SELF and o.i[1] == 4 and o.i[8] == 3]
    l4 = [o for o in ops if o.type == U.OP_CONV and o.i[14] >= 256 and o.i[9] == 3 and o.i[1] == 4]
    shuf = [o for o in ops if o.type == U.OP_CONV and o.flags & 2]
    attn = [o for o in ops if o.type == U.OP_FCONV and o.i[12] == U.FNORM_ATTN]
    n_red = sum(o.type == U.OP_SPLITK_REDUCE for o in ops)
    print(f"B={B} {plan}: {len(ops)} ops; {len(halo)} LDS-tiled 3x3 convs at 32x32 / 16x16, {len(gn_pass)} GroupNorm passes, {len(c4)} fused 4x4 "
          f"GroupNorm-self convs (slices {sorted({o.i[17] for o in c4})}), {len(l4)} LDS-tiled 4x4 convs (groups {sorted({o.i[13] for o in l4})}), "
          f"{n_red} reductions, attention WN {sorted({o.i[16] for o in attn})}")
    assert len(halo) >= 8 and len(gn_pass) >= 8                                      # the 32x32 level (B = 8) / + the 16x16 level (B >= 16) left the fused kernels
    if B >= 32:
        assert len(halo) >= 16
    if plan == "r05":
        assert len(c4) == 16 and all(o.i[17] == 4 and o.i[19] == 0 for o in c4)      # k_conv4_gn_mb: 4 slices, images side by side
        assert n_red >= 8 and all(o.i[14] < 256 for o in shuf)                       # own reductions
    else:
        assert (len(c4), len(l4)) == ((16, 1) if B == 8 else (0, 17)) and all(o.i[13] > 1 for o in l4)      # from 256 rows on: k_conv3_halo_sm, split-K groups
        assert len(shuf) == 3 and all(o.i[14] >= 256 for o in shuf)                  # pixel shuffle on k_conv_lds
    if plan == "r05" or B < 16:
        assert attn and all(o.i[16] == 2 for o in attn)                              # WN = 2 attention projection
    else:                                                                            # from 256 token rows on: LayerNorm pass + LDS-tiled linears + k_attn16
        lin = [o for o in ops if o.type == U.OP_CONV and o.i[1] == 1 and o.i[2] == 16 and o.i[9] == 1]
        assert not attn and sum(o.type == U.OP_ATTN for o in ops) == 5 and len(lin) >= 15 and all(o.i[14] >= 256 for o in lin)
    r, c = rel_err(y, y_ref), cosine(y, y_ref)
    worst = max(rel_err(y[b:b + 1], y_ref[b:b + 1]) for b in range(B))
    print(f"B={B}: rel L2 vs oracle {r:.3e} cosine {c:.6f}; worst image {worst:.3e}")
    assert torch.isfinite(y).all() and r < TOL_REL and c > TOL_COS and worst < TOL_REL
    # the batch plan and the one-image plan agree per image to the same bf16 bound (other tiles, other summation orders)
    y0 = net.forward_with_cond_scale(x[B - 1:].to(DEV), ls[B - 1:].to(DEV), cond_images=cond[B - 1:].to(DEV)).cpu()
    assert rel_err(y0, y[B - 1:]) < TOL_REL


@pytest.mark.parametrize("B", [3, 5, 9, 12, 17, 24])
def test_ragged_batches_agree_with_the_one_image_plan(B):
    """The r06 planner rules switch kernels by row counts (128 / 256 / 1024 / 4096 / 8192 rows, 128-row tiles of whole 4x4 / 8x8 maps, split-K
    groups, k_gemm_rows_ks with 4 | 8 waves, k_gn_one from 256 workgroups): batches that fill their last tile only partly (B = 9: 144 rows at the
    4x4 level = 1.125 tiles; B = 17: the first batch past the 256-row and 4096-row thresholds with a one-map tail; odd B: no k_conv4_gn_mb) are where
    a tail bug would live.  Every image of the batch against the SAME image through the one-image plan (itself pinned to the reference golden):
    same bound as between two plans of one batch, and the batch output is finite everywhere."""
    name = "canonical"
    sd = state(name)
    net = _unet(name, sd)
    g = torch.Generator().manual_seed(700 + B)
    x, cond = torch.randn(B, 4, 32, 32, generator=g), torch.randn(B, 256, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.rand(B, generator=g) * 0.98 + 0.01)
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
    assert torch.isfinite(y).all()
    worst = 0.0
    for b in sorted({0, 1, B // 2, B - 2, B - 1}):
        y1 = net.forward_with_cond_scale(x[b:b + 1].to(DEV), ls[b:b + 1].to(DEV), cond_images=cond[b:b + 1].to(DEV)).cpu()
        worst = max(worst, rel_err(y[b:b + 1], y1))
    print(f"B={B}: worst image vs the one-image plan {worst:.3e}")
    assert worst < TOL_REL


def test_plms_batch8_trajectory_matches_oracle_sampler():
    """51-eval PLMS trajectory at B = 8 (the per-GPU batch of the N = 4 scaling line; first batch on the hybrid plan) against the fp32
    oracle sampler (oracle/unet_ref.plms_sample, pinned to the reference's PLMSSampler by tests/test_oracle_unet.py) with shared noise
    (external/plms.py:54-119): stated trajectory tolerance relative L2 < 3e-2, cosine > 0.9995, per image too."""
    from sparsefusion_amd.vldm import DDPM
    from sparsefusion_amd.plms import PLMSSampler
    name, B, max_thres = "canonical", 8, 0.5
    sd = state(name)
    unet = _unet(name, sd)
    vldm = DDPM(channels=4, unets=(unet,), conditional_encoder=None, conditional_embed_dim=None, image_sizes=(32,),
                timesteps=500, cond_drop_prob=0.1, pred_objectives='noise', conditional=False, auto_normalize_img=False,
                clip_output=True, dynamic_thresholding=False, dynamic_thresholding_percentile=.68, clip_value=10).to(DEV)
    gg = torch.Generator().manual_seed(808)
    lat = 0.5 * torch.randn(B, 4, 32, 32, generator=gg)
    cond = torch.randn(B, 256, 32, 32, generator=gg)
    noises = [torch.randn(B, 4, 32, 32, generator=gg) for _ in range(unet_ref.plms_noise_count(max_thres))]
    with torch.no_grad():
        img_ref, xn_ref, nz_ref, acp_ref, n_evals = unet_ref.plms_sample(lambda z, l: unet_ref.unet_forward(sd, z, l, cond), lat, max_thres, noises)
    assert n_evals == 51
    img, xn, nz, acp = PLMSSampler(vldm, 50).sample(lat.to(DEV), cond_images=cond.to(DEV), use_tqdm=False, return_noise=True,
                                                    max_thres=max_thres, noises=[n.to(DEV) for n in noises])
    assert torch.equal(nz.cpu(), nz_ref) and torch.allclose(xn.cpu(), xn_ref, atol=1e-5)
    rr, cc = rel_err(img.cpu(), img_ref), cosine(img.cpu(), img_ref)
    worst = max(rel_err(img[b:b + 1].cpu(), img_ref[b:b + 1]) for b in range(B))
    print(f"plms canonical B=8 (51 evals) vs oracle sampler: rel {rr:.3e} cos {cc:.6f}; worst image {worst:.3e}")
    assert rr < 3e-2 and cc > 0.9995 and worst < 5e-2 and float(img.abs().max()) <= 10.0


def test_r04_fusions_match_the_r03_plan():
    """The launch fusions of round 4 -- attention core in the prologue of its output projection, statistics slots from k_init_x and
    the Upsample epilogue, NCHW output from the final split-K reduction, the 4x4 res_conv beside the GlobalContext pooling launch --
    against the plan without them (k_attn16, k_slots, k_unpack_out launches, conv1 || res_conv pairs) and, with the GlobalContext pooling back in its own launch, against the oracle: same values to the bf16
    tolerance, fewer ops."""
    name = "canonical"
    sd = state(name)
    net = _unet(name, sd)
    g = torch.Generator().manual_seed(57)
    x, cond = torch.randn(1, 4, 32, 32, generator=g), torch.randn(1, 256, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.tensor([0.45]))
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond)
    ctx = net.begin_sampling(cond.to(DEV), ls.to(DEV))
    y_new = net.eval_prepared(ctx, x.to(DEV), 0).clone().cpu()
    n_new = ctx["plan"].n_body_ops
    net.attn_in_out_proj, net.producer_slots, net.gca_epilogue_pool, net.res_conv_beside_pool = False, False, False, False
    net.drop_plans()
    ctx = net.begin_sampling(cond.to(DEV), ls.to(DEV))
    y_old = net.eval_prepared(ctx, x.to(DEV), 0).clone().cpu()
    n_old = ctx["plan"].n_body_ops
    print(f"body ops {n_old} -> {n_new}; r04 vs r03 plan {rel_err(y_new, y_old):.3e}; vs oracle {rel_err(y_new, y_ref):.3e} / {rel_err(y_old, y_ref):.3e}")
    assert n_new <= n_old - 20
    assert rel_err(y_new, y_old) < TOL_REL and rel_err(y_new, y_ref) < TOL_REL and rel_err(y_old, y_ref) < TOL_REL


@pytest.mark.parametrize("B", [1, 4])
def test_r05_specialised_kernels_match_the_general_kernels(B):
    """The kernels of round 5 -- k_conv4_gn / k_lin4_ln at the 4x4 level, k_layernorm_wave, k_gca_gate_t / k_gca_net0_t, and from B = 2 on
    k_conv4_gn_mb (2 | 4 images per workgroup on one weight slice) and from B = 4 on the own reduction launch of a 4x4 split-K conv1 --
    against the plans behind the planner attributes (`conv4_reduce_min_batch`, `conv4_mb`, then `conv4` / `ln_wave` / `gate_t`: the general kernels): same values to the bf16 tolerance, every plan within the oracle tolerance."""
    name = "canonical"
    sd = state(name)
    net = _unet(name, sd)
    g = torch.Generator().manual_seed(58)
    x, cond = torch.randn(B, 4, 32, 32, generator=g), torch.randn(B, 256, 32, 32, generator=g)
    ls = unet_ref.log_snr(torch.tensor([0.45, 0.9, 0.1, 0.62][:B]))
    with torch.no_grad():
        y_ref = unet_ref.unet_forward(sd, x, ls, cond)
    ys = {}
    switches = ("conv4_reduce_min_batch", "conv4_mb", "conv4", "ln_wave", "gate_t", "conv3s")
    defaults = {k: getattr(net, k) for k in switches}
    for tag, attrs in (("r05", {}), ("conv2_gathers_conv1s_slabs", dict(conv4_reduce_min_batch=0)), ("one_image_per_workgroup", dict(conv4_mb=False)),
                       ("general_conv4_own_reduce", dict(conv4=False)),      # (r06: the pair + own-reduction plan the advisor found broken at B = 4)
                       ("general_pipelined_convs", dict(conv3s=False)),      # r06: k_conv_fused_pipe / _pair / _rc where k_conv3s / k_conv3s_rc run
                       ("general_kernels", dict(conv4_mb=False, conv4=False, ln_wave=False, gate_t=False, conv4_reduce_min_batch=0, conv3s=False))):
        for k in switches:                                   # every variant differs from the default plan by the switches it names only
            setattr(net, k, attrs.get(k, defaults[k]))
        net.drop_plans()
        ys[tag] = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV)).cpu()
        r, c = rel_err(ys[tag], y_ref), cosine(ys[tag], y_ref)
        print(f"B={B} {tag}: rel L2 vs oracle {r:.3e} cosine {c:.6f}; vs the r05 plan {rel_err(ys[tag], ys['r05']):.3e}")
        assert torch.isfinite(ys[tag]).all() and r < TOL_REL and c > TOL_COS and rel_err(ys[tag], ys["r05"]) < TOL_REL


def test_time_table_is_cached_per_schedule_until_the_weights_change():
    """Unet.time_table(log_snrs, key=): the time path is a function of the weights and the schedule only, so a sampler that passes a
    key (PLMSSampler: the tuple of its times) gets the same read-only table on every trajectory; load_state_dict / .to() drop it."""
    name = "small"
    net = _unet(name)
    x, ls, cond = inputs(CONFIGS[name], 2, 3)
    fresh = net.begin_sampling(cond.to(DEV), ls.to(DEV))["table"]
    c1 = net.begin_sampling(cond.to(DEV), ls.to(DEV), table_key=("k", 2))
    c2 = net.begin_sampling(cond.to(DEV), ls.to(DEV), table_key=("k", 2))         # (c1 is stale now: one trajectory per plan)
    assert c1["table"] is c2["table"] and fresh is not c1["table"] and torch.equal(fresh, c1["table"])
    # r05: a loop that draws a new schedule every step (the reference's distillation loop, distillation.py:303) never hits: the
    # cache keeps the two most recent schedules only
    for j in range(5):
        net.begin_sampling(cond.to(DEV), ls.to(DEV), table_key=("other", j))
    assert len(net._table_cache) == 2 and all(k[0][0] == "other" for k in net._table_cache)
    c2 = net.begin_sampling(cond.to(DEV), ls.to(DEV), table_key=("k", 2))
    assert torch.equal(c2["table"], fresh)
    y = net.eval_prepared(c2, x.to(DEV), 1).clone()
    net.load_state_dict(state(name, seed=1), strict=True)              # other weights: the cached table must not survive
    c3 = net.begin_sampling(cond.to(DEV), ls.to(DEV), table_key=("k", 2))
    assert c3["table"] is not c2["table"] and not torch.equal(c3["table"], fresh)
    assert not torch.equal(net.eval_prepared(c3, x.to(DEV), 1), y)


def test_sampler_fast_path_equals_forward():
    """Unet.begin_sampling / eval_prepared (time table once per trajectory + plan body per eval) is the same computation as
    Unet.forward: same kernels on the same operands.  Not bit-identical: the GroupNorm statistics of the 4x4 level meet
    through LDS float atomics whose order varies from launch to launch, and a last-ulp difference there decorrelates the
    bf16 roundings downstream -- two runs of the SAME path differ by the same amount, so the bound is the bf16 one."""
    name = "canonical"
    net = _unet(name)
    g = torch.Generator().manual_seed(77)
    for B in (1, 2):
        x = torch.randn(B, 4, 32, 32, generator=g).to(DEV)
        cond = torch.randn(B, 256, 32, 32, generator=g).to(DEV)
        ls = unet_ref.log_snr(torch.tensor([0.9, 0.5, 0.1, 0.02])).to(DEV)
        for row in (2, 0, 3):
            ctx = net.begin_sampling(cond, ls)                   # forward() below takes the plan over: one trajectory at a time
            y_fast = net.eval_prepared(ctx, x, row).clone()
            y_full = net.forward(x, ls[row].expand(B), cond_images=cond)
            with pytest.raises(RuntimeError, match="stale"):     # the context's latents / time rows lived in that plan's arena
                net.eval_prepared(ctx, x, row)
            assert rel_err(y_fast.cpu(), y_full.cpu()) < TOL_REL, (B, row, rel_err(y_fast.cpu(), y_full.cpu()))
            y_again = net.forward(x, ls[row].expand(B), cond_images=cond)
            print(f"B={B} row={row}: fast vs full {rel_err(y_fast.cpu(), y_full.cpu()):.2e}, full vs full {rel_err(y_again.cpu(), y_full.cpu()):.2e}")


def test_unet_eval_is_bitwise_reproducible():
    """Two evals of the same input give the same BITS on the fully fused plan (the canonical 400 M-parameter configuration:
    every block on k_conv_fused* / k_gca_*): every reduction has a fixed order (GroupNorm statistics by shuffle trees +
    fixed-order LDS sums at 4x4, slot sums elsewhere, split-K slabs reduced in slab order, no float atomics), so a seeded PLMS
    trajectory is reproducible run to run on the same build.  (Plans that fall back to the first-round ops for some layers --
    dim-64 'small' -- still sum their GroupNorm / GlobalContext statistics with f64 / f32 atomics and are not.)"""
    name = "canonical"
    net = _unet(name)
    x, ls, cond = inputs(CONFIGS[name], 2, 5)
    x, ls, cond = x.to(DEV), ls.to(DEV), cond.to(DEV)
    y0 = net.forward(x, ls, cond_images=cond)
    for _ in range(3):
        assert torch.equal(net.forward(x, ls, cond_images=cond), y0)
    ctx = net.begin_sampling(cond, ls[:1].expand(3).contiguous())
    z0 = net.eval_prepared(ctx, x, 1).clone()
    for _ in range(3):
        assert torch.equal(net.eval_prepared(ctx, x, 1), z0)


def test_unet_state_dict_roundtrip_and_errors():
    net = _unet("small")
    sd = net.state_dict()
    assert set(sd.keys()) == set(dict(spec("small")).keys())
    # classifier-free guidance (imagen_pytorch.py:1456-1468): null branch = the image condition zeroed (:1499-1503)
    sdd = state("small")
    x, ls, cond = inputs(CONFIGS["small"], 2, 9)
    with torch.no_grad():
        lo = unet_ref.unet_forward(sdd, x, ls, cond)
        nu = unet_ref.unet_forward(sdd, x, ls, torch.zeros_like(cond))
    y = net.forward_with_cond_scale(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_scale=2.5).cpu()
    assert rel_err(y, nu + (lo - nu) * 2.5) < 2 * TOL_REL
    y1 = net.forward(x.to(DEV), ls.to(DEV), cond_images=cond.to(DEV), cond_drop_prob=1.).cpu()
    assert rel_err(y1, nu) < TOL_REL
    with pytest.raises(RuntimeError):
        net.forward(torch.zeros(1, 4, 32, 32), torch.zeros(1), cond_images=torch.zeros(1, 60, 32, 32))   # CPU tensors


@pytest.mark.parametrize("max_thres,evals", [(0.005, 0), (0.06, 7), (0.995, 51)])
def test_plms_sampler_matches_reference_golden(max_thres, evals):
    from sparsefusion_amd.vldm import DDPM
    from sparsefusion_amd.plms import PLMSSampler
    r = torch.load(f"{GOLD}/plms_sample.pt")[max_thres]
    unet = _unet("small")
    vldm = DDPM(channels=4, unets=(unet,), conditional_encoder=None, conditional_embed_dim=None, image_sizes=(32,),
                timesteps=500, cond_drop_prob=0.1, pred_objectives='noise', conditional=False, auto_normalize_img=False,
                clip_output=True, dynamic_thresholding=False, dynamic_thresholding_percentile=.68, clip_value=10).to(DEV)
    assert set(k for k in vldm.state_dict().keys()) == {"unets.0." + k for k in dict(spec("small")).keys()}
    gg = torch.Generator().manual_seed(r["input_seed"])
    lat = 0.5 * torch.randn(2, 4, 32, 32, generator=gg)
    cond = torch.randn(2, 60, 32, 32, generator=gg)
    torch.manual_seed(r["noise_seed"])
    noises = [torch.randn(2, 4, 32, 32).to(DEV) for _ in range(unet_ref.plms_noise_count(max_thres))]
    calls = [0]
    orig = unet.forward_with_cond_scale

    def counting(*a, **k):
        calls[0] += 1
        return orig(*a, **k)

    unet.forward_with_cond_scale = counting
    orig_prepared = unet.eval_prepared

    def counting_prepared(*a, **k):                      # the sampler's fast path (time table + plan body) is an eval too
        calls[0] += 1
        return orig_prepared(*a, **k)

    unet.eval_prepared = counting_prepared
    img, xn, nz, acp = PLMSSampler(vldm, 50).sample(lat.to(DEV), cond_images=cond.to(DEV), use_tqdm=False, return_noise=True,
                                                    max_thres=max_thres, noises=noises)
    assert calls[0] == evals
    assert torch.equal(nz.cpu(), r["noise"]) and torch.allclose(xn.cpu(), r["x_noisy"], atol=1e-5)
    assert torch.allclose(acp.cpu(), r["alpha_cumprod"], atol=1e-6)
    rr, cc = rel_err(img.cpu(), r["img"]), cosine(img.cpu(), r["img"])
    print(f"plms[{max_thres}] rel {rr:.3e} cos {cc:.6f}")
    assert rr < (1e-6 if evals == 0 else 3e-2) and cc > 0.9995       # measured 1.0e-3 (7 evals) / 1.8e-2 (51 evals, dim-64 config)
    assert float(img.abs().max()) <= 10.0


def test_plms_canonical_config_trajectory_matches_reference_golden():
    """The headline configuration end to end: the reference's own PLMSSampler.sample on the 400.68 M-parameter UNet at
    max_thres = 0.5 (50 steps = 51 evals, B = 1: sparsefusion/distillation.py:304, external/plms.py:54-119), golden from
    tests/golden/make_golden_unet.py --canonical-plms.  51 chained bf16-operand evals: stated trajectory tolerance
    relative L2 < 3e-2 and cosine > 0.9995 (measured 4e-3 ... 1.5e-2 / 0.9999: a 2x regression fails)."""
    from sparsefusion_amd.vldm import DDPM
    from sparsefusion_amd.plms import PLMSSampler
    r = torch.load(f"{GOLD}/plms_sample_canonical.pt")
    unet = _unet("canonical")
    vldm = DDPM(channels=4, unets=(unet,), conditional_encoder=None, conditional_embed_dim=None, image_sizes=(32,),
                timesteps=500, cond_drop_prob=0.1, pred_objectives='noise', conditional=False, auto_normalize_img=False,
                clip_output=True, dynamic_thresholding=False, dynamic_thresholding_percentile=.68, clip_value=10).to(DEV)
    gg = torch.Generator().manual_seed(r["input_seed"])
    lat = 0.5 * torch.randn(1, 4, 32, 32, generator=gg)
    cond = torch.randn(1, 256, 32, 32, generator=gg)
    torch.manual_seed(r["noise_seed"])
    noises = [torch.randn(1, 4, 32, 32).to(DEV) for _ in range(unet_ref.plms_noise_count(r["max_thres"]))]
    calls = [0]
    orig = unet.eval_prepared

    def counting(*a, **k):
        calls[0] += 1
        return orig(*a, **k)

    unet.eval_prepared = counting
    img, xn, nz, acp = PLMSSampler(vldm, 50).sample(lat.to(DEV), cond_images=cond.to(DEV), use_tqdm=False, return_noise=True,
                                                    max_thres=r["max_thres"], noises=noises)
    assert calls[0] == 51
    assert torch.equal(nz.cpu(), r["noise"]) and torch.allclose(xn.cpu(), r["x_noisy"], atol=1e-5)
    assert torch.allclose(acp.cpu(), r["alpha_cumprod"], atol=1e-6)
    rr, cc = rel_err(img.cpu(), r["img"]), cosine(img.cpu(), r["img"])
    print(f"plms canonical (51 evals) rel {rr:.3e} cos {cc:.6f}")
    assert rr < 3e-2 and cc > 0.9995 and float(img.abs().max()) <= 10.0


def test_plms_internal_draws_match_injected_ones():
    """Without injected noises the sampler draws the reference's number of gaussians (1 + 2n + 2) in ONE randn of the
    whole trajectory and uses them in the reference's roles: injecting the same tensor slice by slice is the same run."""
    from sparsefusion_amd.vldm import DDPM
    from sparsefusion_amd.plms import PLMSSampler
    unet = _unet("small")
    vldm = DDPM(channels=4, unets=(unet,), image_sizes=(32,), timesteps=500, conditional=False, clip_output=True,
                dynamic_thresholding=False, clip_value=10).to(DEV)
    lat, cond = torch.randn(1, 4, 32, 32, device=DEV), torch.randn(1, 60, 32, 32, device=DEV)
    torch.manual_seed(5)
    noises = list(torch.randn(unet_ref.plms_noise_count(0.04), 1, 4, 32, 32, device=DEV).unbind(0))
    a = PLMSSampler(vldm, 50).sample(lat, cond_images=cond, use_tqdm=False, return_noise=True, max_thres=0.04, noises=noises)
    torch.manual_seed(5)
    b = PLMSSampler(vldm, 50).sample(lat, cond_images=cond, use_tqdm=False, return_noise=True, max_thres=0.04)
    assert torch.equal(a[2], b[2]) and rel_err(a[0].cpu(), b[0].cpu()) < 2e-2     # eval-to-eval bf16 decorrelation only
