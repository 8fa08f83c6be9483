"""Planner invariants checked WITHOUT a GPU: every model's static launch plan is built in sizing mode (fake device
pointers) and its op list is audited -- operand presence, alignment contracts of the kernels, op-code agreement with the
C header, lazy tensors all consumed.  Guards the host logic of unet.py / vae.py / lpips.py / eft.py."""
import re
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = torch.device("cpu")


def _header_enum():
    txt = open(os.path.join(ROOT, "include", "sparsefusion_hip.h")).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"(SF_OP_[A-Z_0-9]+)\s*=\s*(\d+)", txt)}


def test_op_codes_match_header():
    from sparsefusion_amd import unet, lpips, eft
    e = _header_enum()
    assert (unet.OP_CONV, unet.OP_GN_ACT, unet.OP_LN, unet.OP_GEMV, unet.OP_ATTN, unet.OP_GCA_POOL, unet.OP_ELTWISE, unet.OP_MEMSET,
            unet.OP_TIME_EMB, unet.OP_SPLITK_REDUCE) == tuple(e[k] for k in (
                "SF_OP_CONV", "SF_OP_GN_ACT", "SF_OP_LN", "SF_OP_GEMV", "SF_OP_ATTN", "SF_OP_GCA_POOL", "SF_OP_ELTWISE", "SF_OP_MEMSET",
                "SF_OP_TIME_EMB", "SF_OP_SPLITK_REDUCE"))
    assert (lpips.OP_POOL, lpips.OP_LPIPS, eft.OP_POOL, eft.OP_EFT) == (e["SF_OP_POOL"], e["SF_OP_LPIPS"], e["SF_OP_POOL"], e["SF_OP_EFT"])
    assert sorted(e.values()) == list(range(1, len(e) + 1))                      # dense, no duplicates


def _audit(ops, name, sized=False):
    from sparsefusion_amd import unet
    n_conv = 0
    for kk, o in enumerate(ops):
        where = f"{name} op {kk} type {o.type}"
        if o.type == unet.OP_CONV:
            n_conv += 1
            B, H, W, Cin, Ho, Wo, Cout, ldc, co_off, kh, kw, stride, pad, groups, tile = list(o.i)[:15]
            assert o.p[0] and o.p[1] and o.p[3], where
            assert Cin % 32 == 0 and Cin > 0 and Cout > 0 and ldc >= co_off + (Cout if not o.flags & 2 else Cout // 4), where
            assert groups >= 1 and kh == kw and stride in (1, 2), where
            if groups > 1:
                assert o.p[5] or not sized, where + ": split-K without workspace"
                assert not (o.flags & 2), where + ": pixel shuffle cannot be split-K"
            if tile >= 256:                                                     # k_conv_lds / k_conv_glds contract (r06: split-K groups, pixel shuffle)
                assert tile - 256 in (4, 8) and 1 <= groups <= 16, where
                assert groups == 1 or not (o.flags & 128), where + ": no GroupNorm partials under split-K"
                if o.flags & 2:
                    assert groups == 1 and Cout % 4 == 0 and not (o.flags & (4 | 8 | 32 | 64 | 128)) and not o.p[4], where
                stages = (kh * kw * (Cin // 32) + 1) // 2
                assert groups <= stages, where + ": more split-K groups than stages"
            else:
                assert tile // 16 in (1, 2, 4) and tile % 16 in (1, 2, 4), where
            if o.flags & 8:
                assert groups > 1 and not (o.flags & (4 | 32 | 64)), where + ": deferred reduce only for plain split-K"
        elif o.type == unet.OP_GN_ACT:
            B, HW, C1, C2, _, lazy, lgroups, npad, G = list(o.i)[:9]
            G = G or 8
            assert o.p[0] and o.p[2] and o.p[3] and o.p[5] and o.p[7], where
            assert (C1 + C2) % (4 * G) == 0 and C1 % 4 == 0, where
            if lazy == 1:
                assert (o.p[8] or not sized) and lgroups >= 1 and npad % 4 == 0, where       # p[8] = the split-K workspace
            if lazy == 2:
                assert o.p[8] and o.p[9], where
        elif o.type == unet.OP_FCONV:
            n_conv += 1
            B, H, W, C1, C2, Cout, ldc, co_off, k, lmode, lgroups, npad, norm, G, TR, WM, WN, S = list(o.i)[:18]
            C = C1 + C2
            second_of_pair = kk > 0 and ops[kk - 1].type == unet.OP_FCONV and (ops[kk - 1].flags & 16)
            # the res_conv half of a pair reads a lazy source without materialising it (p[0] null): only there
            assert (o.p[0] or (second_of_pair and lmode)) and o.p[7] and (o.p[9] or S > 1), where
            if (o.flags & 16) and ops[kk + 1].type == unet.OP_GCA:     # res_conv beside the pooling launch (k_gca_pool_rc)
                nx = ops[kk + 1]
                assert nx.flags == 1 and norm == unet.FNORM_NONE and k == 1 and [WM, WN, S] == [1, 1, 1] and not lmode, where + ": pool pair"
                assert nx.i[0] == B * H * W and nx.i[1] == Cout, where + ": pool pair halves belong to one block"
            elif o.flags & 16:
                nx = ops[kk + 1]
                assert nx.type == unet.OP_FCONV and nx.i[12] == unet.FNORM_NONE and list(nx.i)[15:18] == [WM, WN, 1] and nx.i[9] == lmode, where + ": pair"
                assert (WM, WN, norm) in unet.PAIR_TILES and not (nx.flags & 16), where + ": pair variant"
            if second_of_pair:
                assert norm == unet.FNORM_NONE and list(o.p)[1:4] == list(ops[kk - 1].p)[1:4], where + ": pair halves read the same source"
            assert C % 32 == 0 and C1 % 32 == 0 and k in (1, 3) and TR * W == 16 * WM and H % TR == 0, where
            assert (C // 32) % S == 0 and WM in (1, 2, 4) and WN in (1, 2), where
            if S > 1:
                assert (o.p[11] or not sized) and not (o.flags & 4) and not o.p[12], where + ": sliced convs write slabs only"
            if norm in (unet.FNORM_GN_SELF, unet.FNORM_GN_SLOTS):
                Cg = C // G
                assert o.p[13] and o.p[14] and (C // S) % Cg == 0 and (C // S) // Cg <= 8, where
                if norm == unet.FNORM_GN_SELF:
                    assert TR == H, where
                else:
                    assert o.p[4] and (o.p[6] or not C2) and Cg % 16 == 0 and lmode != 1, where + ": slot statistics"
            if lmode:
                assert (o.p[1] or not sized) and (lmode == 1 or (o.p[2] and o.p[3])), where
                assert not o.p[0] or (o.p[3] != o.p[0] and o.p[1] != o.p[0]), where
            if lmode == 1 and S > 1 and sized:
                assert o.p[1] != o.p[11], where + ": a conv must not overwrite the slabs it reads"
            h = k // 2
            Cs = C // S
            stride = Cs * 2 + ((32 - (Cs * 2) % 256) + 256) % 256
            if o.flags & 32:                                                    # k_conv_fused_pipe contract
                assert norm == unet.FNORM_GN_SLOTS and k == 3 and S == 1 and lmode == 0 and C % 128 == 0 and G == 8, where
                assert (WM, WN, (TR + 2) * W // 8) in unet.PIPE_TILES and ((TR + 2) * W) % 8 == 0, where
                buf = (((TR + 2) * (W + 2) + 1) * 288 + 15) // 16 * 16
                assert 2 * buf + 4096 * WM * WN + 8 * C + 2688 <= unet.LDS_MAX, where
            else:
                assert ((TR + 2 * h) * (W + 2 * h) + 1) * stride + 8192 * WM * WN + 8 * Cs + 2688 <= unet.LDS_MAX, where
        elif o.type == unet.OP_SLOTS:
            assert (o.p[0] or o.p[5] or not sized) and o.p[4] and o.i[0] % 16 == 0 and o.i[1] % 16 == 0 and (not o.p[1] or (o.p[2] and o.p[3])), where
            if o.i[3]:                                                        # split-K source: slabs + output buffer
                assert (o.p[5] or not sized) and o.p[3] and not o.p[1] and o.i[4] % 4 == 0 and o.i[4] >= o.i[1], where
        elif o.type == unet.OP_LN:
            assert o.p[0] and o.p[1] and o.p[3] and o.i[1] % 64 == 0 and o.i[1] <= 2048, where
        elif o.type == unet.OP_GEMV:
            assert o.p[0] and o.p[1] and o.p[3] and 1 <= o.i[0] <= 64 and o.i[3] % 8 == 0, where      # (<= 8 rows: k_gemv; 9 .. 64: k_gemm_rows / _ks)
    return n_conv


def test_unet_plan_invariants():
    from sparsefusion_amd import unet as unet_mod
    from sparsefusion_amd.unet import Unet, _Plan
    net = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
               layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    for fused in (False, True):
        net.fused = fused
        for B in (1, 4):
            for lazy in (0, 3):
                net.lazy_consumers = lazy
                plan = _Plan(net, B, CPU).build()
                # conv / linear layers per eval; the fused plan merges the q and k|v projections of its 3 self-attentions
                assert _audit(plan.ops, f"unet B={B} lazy={lazy} fused={fused}") == (91 if fused else 94)      # Parallel(3x3, 1x1) of the last level = one merged conv
                assert plan.ws_owner is None or plan.ws_owner.lazy is None             # nothing left un-materialised
                assert plan.zero.off >= 0 and plan.misc.off > plan.zero.off
                if fused:                                                              # GroupNorm lives inside the convs now
                    assert sum(o.type == unet_mod.OP_GN_ACT for o in plan.ops) == 0
                    assert sum(o.type == unet_mod.OP_FCONV for o in plan.ops) >= 54
    # second pass with real arenas (host memory stands in for HBM): pointers are now real, the workspaces exist
    net.lazy_consumers = 3
    s = _Plan(net, 1, CPU).build()
    sized = _Plan(net, 1, CPU, (s.zero.off, s.misc.off + s.ws_bytes + s.ws2_bytes + 512, s.ws_bytes, s.ws2_bytes)).build()
    assert _audit(sized.ops, "unet sized", sized=True) == 91 and len(sized.ops) == len(s.ops)
    lo, hi = sized.misc.buf.data_ptr(), sized.misc.buf.data_ptr() + sized.misc.buf.numel()
    for o in sized.ops:                                                            # every activation operand lies inside an arena
        if o.type == 1:
            assert lo <= o.p[3] < hi and lo <= o.p[0] < hi
        if o.type == unet_mod.OP_FCONV:
            assert (not o.p[0] or lo <= o.p[0] < hi) and (lo <= o.p[9] < hi) and (not o.p[11] or lo <= o.p[11] < hi)
    # the lazy plan drops launches but never changes the set of convs
    net.fused = False
    net.lazy_consumers = 3
    lazy_ops = len(_Plan(net, 1, CPU).build().ops)
    net.lazy_consumers = 0
    assert lazy_ops < len(_Plan(net, 1, CPU).build().ops)


def test_time_table_plan_uses_one_launch_per_linear():
    """A sampler's time table (Unet.time_table, 51 log-snr rows): every Linear of the time path is ONE OP_GEMV with all rows
    (k_gemm_rows: <= 64 rows on the MFMA M side), not one per 8 rows."""
    from sparsefusion_amd import unet as unet_mod
    from sparsefusion_amd.unet import Unet, _TimePlan
    net = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
               layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    plan = _TimePlan(net, 51, CPU).build()
    gemvs = [o for o in plan.ops if o.type == unet_mod.OP_GEMV]
    assert gemvs and all(o.i[0] == 51 and o.i[3] % 8 == 0 and o.p[1] for o in gemvs)
    assert max(o.i[1] for o in gemvs) == net.ss_total                 # the 27 time_mlp Linears batched into one matrix
    big = _TimePlan(net, 100, CPU).build()                            # more rows than one launch takes: chunks of 64
    assert sorted({o.i[0] for o in big.ops if o.type == unet_mod.OP_GEMV}) == [36, 64]


def test_vae_lpips_eft_plan_invariants():
    from sparsefusion_amd.vae import AutoencoderKL, _VaePlan
    from sparsefusion_amd.lpips import LPIPS, _LpipsPlan
    from sparsefusion_amd.eft import EpipolarFeatureTransformer, _EftPlan
    vae = AutoencoderKL()
    assert _audit(_VaePlan(vae, "enc", 1, CPU).build().ops, "vae enc") == 34
    assert _audit(_VaePlan(vae, "dec", 2, CPU).build().ops, "vae dec") == 42 + 2       # two per-sample attention GEMM pairs more
    lp = LPIPS()
    fwd = _LpipsPlan(lp, 1, 256, CPU).build_forward()
    assert _audit(fwd.ops, "lpips fwd") == 13
    bwd = _LpipsPlan(lp, 1, 256, CPU, fwd=fwd).build_backward()
    assert _audit(bwd.ops, "lpips bwd") == 13
    eft = EpipolarFeatureTransformer(use_r=True, encoder='resnet18', return_features=True, remove_unused_layers=False)
    enc = _EftPlan(eft, 6, CPU).build_encoder(6, 256)
    assert _audit(enc.ops, "eft enc") == 1 + 2 * 6 + 2                              # conv1, 6 BasicBlocks, 2 downsample convs
    f = _EftPlan(eft, 6, CPU)
    f.images_ptr = 1
    f.build_forward(6, 1024, 20, enc, 256)
    assert _audit(f.ops, "eft fwd") == 3 * (1 + 4 * 4)                              # 3 x (pre + 4 layers x (qkv, out, ff1, ff2))


def test_vae_gn_epilogue_plan(monkeypatch):
    """Every GroupNorm whose input was last written by a whole-tensor k_conv_lds launch takes its statistics from that conv's
    epilogue (flag 128 + partials buffer + group width on the conv, OP_GN_FINALIZE, flag 2 on the GroupNorm); the others keep
    the statistics pass.  Default since r03 (measured on the GPU); `vae.gn_epilogue = False` restores the statistics pass everywhere."""
    from sparsefusion_amd import unet as unet_mod
    from sparsefusion_amd.vae import AutoencoderKL, _VaePlan
    off = AutoencoderKL()
    off.gn_epilogue = False
    base = _VaePlan(off, "dec", 1, CPU).build()
    assert not any(o.type == unet_mod.OP_GN_FINALIZE or (o.type == unet_mod.OP_CONV and o.flags & 128) for o in base.ops)
    vae = AutoencoderKL()
    for kind, B in (("enc", 1), ("dec", 2)):
        s = _VaePlan(vae, kind, B, CPU).build()
        plan = _VaePlan(vae, kind, B, CPU, (s.zero.off, s.misc.off + 512, 0, 0)).build()
        ops = plan.ops
        gns = [k for k, o in enumerate(ops) if o.type == unet_mod.OP_GN_ACT]
        ready = [k for k in gns if ops[k].flags & 2]
        assert len(ready) >= len(gns) // 2, (kind, len(ready), len(gns))                # the large maps (the small ones run k_conv_igemm)
        for k in ready:
            fin = ops[k - 1]
            assert fin.type == unet_mod.OP_GN_FINALIZE and fin.p[1] == ops[k].p[7]                       # same statistics buffer
            B_, HW, C = ops[k].i[0], ops[k].i[1], ops[k].i[2]
            assert fin.i[0] == B_ and fin.i[1] == HW // 128 and fin.i[2] == 32 and HW % 128 == 0
            prod = [o for o in ops[:k] if o.type == unet_mod.OP_CONV and o.p[3] == ops[k].p[0]][-1]      # last writer of the input
            assert prod.flags & 128 and prod.p[6] == fin.p[0] and prod.i[15] == C // 32 and prod.i[14] >= 256
            assert prod.i[6] == prod.i[7] == C and prod.i[8] == 0
        assert sum(1 for o in ops if o.type == unet_mod.OP_CONV and o.flags & 128) == len(ready)


def test_lds_conv_selection_rule():
    """Large-M layers switch to the LDS-tiled kernel by tile count; small-M UNet layers at B = 1 never do."""
    from sparsefusion_amd.unet import OP_CONV, Unet, _Plan
    from sparsefusion_amd.vae import AutoencoderKL, _VaePlan
    net = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
               layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    net.fused = False                                                             # the first-round plan (fused convs never use it)
    assert all(o.i[14] < 256 for o in _Plan(net, 1, CPU).build().ops if o.type == OP_CONV)
    assert any(o.i[14] >= 256 for o in _Plan(net, 16, CPU).build().ops if o.type == OP_CONV)
    dec = _VaePlan(AutoencoderKL(), "dec", 1, CPU).build()
    big = [o for o in dec.ops if o.type == OP_CONV and o.i[4] >= 128]
    assert big and all(o.i[14] >= 256 or o.i[6] < 64 for o in big)                 # every wide 128^2 / 256^2 conv is LDS-tiled


def test_vae_upsample_reads_the_operand_twin():
    """r03: the decoder's three Upsample convs read an operand-type twin of the block output that the producing conv2 (an LDS-tiled
    kernel from 64 tiles on when k_conv3_halo can take the layer, csrc/conv_halo.h) writes in its epilogue -- bf16 A operand, the
    nearest-x2 view folded into the halo addressing -- instead of the fp32 tensor; `vae.conv_twin = False` restores the fp32 read."""
    from sparsefusion_amd.unet import OP_CONV
    from sparsefusion_amd.vae import AutoencoderKL, _VaePlan
    ops = _VaePlan(AutoencoderKL(), "dec", 1, CPU).build().ops
    ups = [k for k, o in enumerate(ops) if o.type == OP_CONV and o.flags & 16]
    assert len(ups) == 3
    for k in ups:
        o = ops[k]
        assert not (o.flags & 1) and o.i[14] >= 256                                             # operand-type input, LDS-tiled kernel
        prod = [q for q in ops[:k] if q.type == OP_CONV and q.p[5] == o.p[0]]
        assert len(prod) == 1 and prod[0].i[14] >= 256 and prod[0].i[13] == 1                   # written by one LDS-tiled conv (no split-K workspace)
        assert prod[0].i[6] == prod[0].i[7] == o.i[3] and prod[0].i[8] == 0                     # dense [pixel][Cout] = the consumer's Cin
        assert prod[0].i[4] * 2 == o.i[1] and prod[0].i[5] * 2 == o.i[2]                        # consumer dims are the upsampled view
    assert sum(1 for q in ops if q.type == OP_CONV and q.i[14] >= 256 and q.p[5]) == 3 + 2     # + the Upsample convs' own twins for the two nin_shortcuts
    nin = [o for o in ops if o.type == OP_CONV and o.i[9] == 1 and o.i[3] != o.i[6] and o.i[1] >= 128]
    assert len(nin) == 2 and all(not (o.flags & 1) and o.i[14] >= 256 for o in nin)              # 1x1 shortcuts on the operand twin
    enc = _VaePlan(AutoencoderKL(), "enc", 1, CPU).build().ops
    down = [o for o in enc if o.type == OP_CONV and o.i[11] == 2]
    assert len(down) == 3 and all(not (o.flags & 1) for o in down)                              # Downsample convs read the block's twin
    assert all(any(q.type == OP_CONV and q.p[5] == o.p[0] for q in enc) for o in down)
    assert all(q.i[14] >= 256 for q in ops if q.type == OP_CONV and q.i[9] == 3 and q.i[4] == 32 and q.i[3] == 512 and q.i[6] == 512)   # 32x32 layers: 64 tiles
    plain = AutoencoderKL()
    plain.conv_twin = False
    ops0 = _VaePlan(plain, "dec", 1, CPU).build().ops
    assert all(o.flags & 1 for o in ops0 if o.type == OP_CONV and o.flags & 16)
    assert not any(q.p[5] for q in ops0 if q.type == OP_CONV and q.i[14] >= 256)


def test_every_planned_fused_conv_has_a_kernel_variant():
    """The planner (unet.py) and the instantiation lists (csrc/fused_host.h SF_FCONV_*_VARIANTS) are two tables: every OP_FCONV of
    the canonical plans at B = 1 .. 32 -- including the batch sizes no GPU test runs -- must name an instantiated
    (WM, WN, norm, lazy) / (WM, WN, EPT) combination (r04: the attention output projection at B >= 8 planned WN = 2, which had none)."""
    from sparsefusion_amd import unet as U
    txt = open(os.path.join(ROOT, "sparsefusion_amd", "csrc", "fused_host.h")).read()
    norms = {"FNORM_NONE": 0, "FNORM_GN_SELF": 1, "FNORM_GN_SLOTS": 2, "FNORM_LN": 3, "FNORM_ATTN": 4}

    def table(macro):
        body = re.search(r"#define " + macro + r"\(X\)((?:\s*\\\n\s*X\([^)]*\))+)", txt).group(1)
        return [tuple(t.strip() for t in m.split(",")) for m in re.findall(r"X\(([^)]*)\)", body)]

    plain = {(int(a), int(b), norms[n], int(lz)) for a, b, _, n, lz in table("SF_FCONV_VARIANTS")}
    pairs = {(int(a), int(b), norms[n], int(lz)) for a, b, _, n, lz in table("SF_FCONV_PAIR_VARIANTS")}
    pipes = {(int(a), int(b), int(e)) for a, b, e in table("SF_FCONV_PIPE_VARIANTS")}
    net = U.Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    for B in (1, 2, 3, 4, 8, 16, 32):
        ops = U._Plan(net, B, CPU).build().ops
        for k, o in enumerate(ops):
            if o.type != U.OP_FCONV:
                continue
            WM, WN, norm, lazy, TR, W = o.i[15], o.i[16], o.i[12], o.i[9], o.i[14], o.i[2]
            second = k > 0 and ops[k - 1].type == U.OP_FCONV and (ops[k - 1].flags & 16)
            if o.flags & 32:
                assert (WM, WN, (TR + 2) * W // 8) in pipes, (B, k, WM, WN, TR, W)
            elif (o.flags & 16) and ops[k + 1].type == U.OP_GCA:
                assert (WM, WN, norm, lazy) == (1, 1, 0, 0), (B, k)  # k_gca_pool_rc<1, 1, 12> (unet_fused.hip::run_pool_rc_pair)
            elif o.flags & 16:
                assert (WM, WN, norm, lazy) in pairs, (B, k, WM, WN, norm, lazy)
            elif second:
                assert norm == 0                                  # the res_conv half runs inside its partner's instantiation
            else:
                assert (WM, WN, norm, lazy) in plain, (B, k, WM, WN, norm, lazy)


def test_conv3s_tables_agree_and_the_b1_plan_uses_them():
    """r06: the planner's CONV3S_VARIANTS and csrc/fused_host.h's SF_CONV3S_VARIANTS are one table; every variant has a GPU op case
    (tests/fused_cases.py), with and without epilogue pooling; the canonical B = 1 plan sends its 29 single-source slot-GroupNorm 3x3 convs
    there with the planned tile widths."""
    import fused_cases as fc
    from sparsefusion_amd import unet as U
    txt = open(os.path.join(ROOT, "sparsefusion_amd", "csrc", "fused_host.h")).read()
    body = re.search(r"#define SF_CONV3S_VARIANTS\(X\)((?:\s*\\\n(?:\s*X\([^)]*\))+)+)", txt).group(1)
    host = {tuple(int(v) for v in m.split(",")) for m in re.findall(r"X\(([^)]*)\)", body)}
    assert host == U.CONV3S_VARIANTS
    body = re.search(r"#define SF_CONV3S_RC_VARIANTS\(X\)((?:\s*\\\n(?:\s*X\([^)]*\))+)+)", txt).group(1)
    host_rc = {tuple(int(v) for v in m.split(",")) for m in re.findall(r"X\(([^)]*)\)", body)}
    assert host_rc == U.CONV3S_RC_VARIANTS
    covered_rc = {(kw["H"].bit_length() - 1, kw["C1"], kw["C2"], kw["Cout"], kw["tw"].bit_length() - 1, kw["WM"], kw["WN"])
                  for name, kw in fc.CONV_CASES_FULL.items() if name.startswith("conv3s_rc_")}
    assert host_rc <= covered_rc, host_rc - covered_rc
    covered = set()
    for name, kw in fc.CONV_CASES_FULL.items():
        if name.startswith("conv3s_") and not name.startswith("conv3s_rc_") and not kw.get("keep_pipe"):
            covered.add(((kw["H"].bit_length() - 1, kw["C1"], kw["tw"].bit_length() - 1, kw["WM"], kw["WN"]), bool(kw.get("pool"))))
    for v in host:
        assert (v, False) in covered, f"k_conv3s variant {v} has no GPU op case"
        assert (v, True) in covered, f"k_conv3s variant {v} (POOL) has no GPU op case"
    net = U.Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    ops = U._Plan(net, 1, CPU).build().ops
    pipe = [o for o in ops if o.type == U.OP_FCONV and (o.flags & 32) and not (o.flags & 16)]
    assert len(pipe) == 29
    taken = 0
    for o in pipe:
        H, C, WM, WN, code, pool = o.i[1], o.i[3], o.i[15], o.i[16], o.i[19], bool(o.flags & 64)
        tw = (code >> 2) or H
        assert not (code & 2) and o.i[4] == 0 and o.i[5] == C
        assert (H.bit_length() - 1, C, tw.bit_length() - 1, WM, WN) in host, (H, C, tw, WM, WN)
        assert tw == {32: net.conv3s_tw32, 16: net.conv3s_tw16, 8: net.conv3s_tw8}[H] or tw == H
        taken += 1
    assert taken == 29
    pairs = [o for o in ops if o.type == U.OP_FCONV and (o.flags & 32) and (o.flags & 16)]
    assert len(pairs) == 9                                         # conv1 + res_conv of the blocks that change width: all on k_conv3s_rc
    for o in pairs:
        H, tw = o.i[1], (o.i[19] >> 2) or o.i[1]
        assert (H.bit_length() - 1, o.i[3], o.i[4], o.i[5], tw.bit_length() - 1, o.i[15], o.i[16]) in host_rc, (H, o.i[3], o.i[4], o.i[5], tw)
    net.conv3s = False
    assert all(o.i[19] & 2 for o in U._Plan(net, 1, CPU).build().ops if o.type == U.OP_FCONV and (o.flags & 32) and not (o.flags & 16))


def test_a_paired_conv_is_followed_by_its_partner():
    """A conv emitted with flag 16 (first half of a pair) shares its launch with the NEXT op: sf_plan_fused_pair takes ops[k + 1], which
    must be the res_conv (OP_FCONV) or the pooling launch (OP_GCA).  r05's own-launch split-K reduction once stood between the two
    (B = 5, 7 by default; B = 4 .. 6 with conv4 = False): every batch size 1 .. 9 under every planner switch that shapes the 4x4 level."""
    from sparsefusion_amd import unet as U
    net = U.Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    switches = ("conv4", "res_conv_beside_pool", "conv4_mb", "pair_res_conv")
    defaults = {k: getattr(net, k, True) for k in switches}
    n_pairs = 0
    for attrs in ({}, {"conv4": False}, {"res_conv_beside_pool": False}, {"conv4_mb": False}, {"conv4": False, "res_conv_beside_pool": False}):
        for k in switches:
            setattr(net, k, attrs.get(k, defaults[k]))
        for B in range(1, 10):
            ops = U._Plan(net, B, CPU).build().ops
            for k, o in enumerate(ops):
                if o.type == U.OP_FCONV and o.flags & 16:
                    n_pairs += 1
                    assert ops[k + 1].type in (U.OP_FCONV, U.OP_GCA), (attrs, B, k, ops[k + 1].type)
                    if ops[k + 1].type == U.OP_FCONV:
                        assert ops[k + 1].i[8] == 1 and ops[k + 1].i[12] == U.FNORM_NONE, (attrs, B, k)      # the 1x1 res_conv
    assert n_pairs > 100


def test_4x4_level_rules_for_more_than_one_image():
    """r05: the 4x4 GroupNorm-self convs keep the 4-slice geometry of k_conv4_gn / k_conv4_gn_mb up to B = 4 and at every even
    B (the workgroup-count rule alone gives 2 slices at B = 2 and 1 from B = 4 on); from B = 4 on a split-K conv1 is reduced by its own
    launch, so no 4x4 conv gathers another 4x4 conv's slabs; the planner attributes restore the older plans, `conv4_mb = False` marks the ops
    (i[19] bit 0: one image per workgroup) and leaves the slice count to the workgroup rule beyond B = 4."""
    from sparsefusion_amd import unet as U
    net = U.Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    net.unfused_min_rows_4 = 0                  # (r06: from B = 8 on the default plan takes the 4x4 ResnetBlocks off the fused kernels, see the next test)

    def convs4(B):
        ops = U._Plan(net, B, CPU).build().ops
        c4 = [o for o in ops if o.type == U.OP_FCONV and o.i[12] == U.FNORM_GN_SELF and o.i[1] == 4 and o.i[8] == 3]
        return ops, c4

    ops1, c1 = convs4(1)
    assert len(c1) == 16 and all(o.i[17] == 4 and o.i[19] == 0 for o in c1)
    n_red1 = sum(o.type == U.OP_SPLITK_REDUCE for o in ops1)
    for B in (2, 4, 8, 32):
        ops, c4 = convs4(B)
        assert len(c4) == 16 and all(o.i[17] == 4 and o.i[19] == 0 for o in c4), B
        n_red = sum(o.type == U.OP_SPLITK_REDUCE for o in ops)
        lazy_splitk = sum(o.i[9] == 1 for o in c4)
        if B >= 4:
            assert n_red >= n_red1 + 8 and lazy_splitk < sum(o.i[9] == 1 for o in c1), (B, n_red, n_red1, lazy_splitk)
        else:
            assert n_red == n_red1, (B, n_red, n_red1)
    assert all(o.i[17] == 4 for o in convs4(3)[1]) and all(o.i[17] < 4 for o in convs4(5)[1])        # odd B: up to conv4_slices_max_batch only
    net.conv4_mb = False
    _, c8 = convs4(8)
    assert all(o.i[17] < 4 and o.i[19] == 1 for o in c8)                      # (1 slice; 2 where a 2048-channel frame needs them)
    _, c4 = convs4(4)
    assert all(o.i[17] == 4 and o.i[19] == 1 for o in c4)
    net.conv4_mb, net.conv4_reduce_min_batch = True, 0
    ops4, c4 = convs4(4)
    assert sum(o.type == U.OP_SPLITK_REDUCE for o in ops4) == n_red1 and sum(o.i[9] == 1 for o in c4) == sum(o.i[9] == 1 for o in c1)


def test_large_batch_plans_run_the_4x4_level_on_the_lds_tiled_kernels():
    """r06: from B = 8 on (lds_mid_min_batch) convs of >= 128 rows that have too few 128-row tiles run on the LDS-tiled kernels with split-K
    groups, and the Upsample 1x1 convs run the pixel shuffle there; from B = 16 on (unfused_min_rows_4 = 256 rows) the 4x4 ResnetBlocks run
    GroupNorm as its own pass and their 3x3 convs on k_conv3_halo_sm (operand-type input, groups <= Cin / 128: two 64-channel chunks per
    group at least, workgroups <= 256 + one tile row); B <= 4 plans hold no such op; the switches restore the r05 plans."""
    from sparsefusion_amd import unet as U
    net = U.Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    for B in (1, 2, 4):
        ops = U._Plan(net, B, CPU).build().ops
        assert not any(o.type == U.OP_CONV and o.i[14] >= 256 and (o.i[13] > 1 or o.flags & 2) for o in ops), B
    for B in (8, 16, 32):
        ops = U._Plan(net, B, CPU).build().ops
        c4 = [o for o in ops if o.type == U.OP_CONV and o.i[1] == 4 and o.i[9] == 3 and o.i[11] == 1]
        fused4 = sum(o.type == U.OP_FCONV and o.i[12] == U.FNORM_GN_SELF and o.i[1] == 4 and o.i[8] == 3 for o in ops)
        assert all(o.i[14] >= 256 and o.i[13] > 1 for o in c4), (B, [(o.i[14], o.i[13]) for o in c4])
        assert (len(c4), fused4) == ((1, 16) if B == 8 else (17, 0)), (B, len(c4), fused4)
        for o in c4:
            bnf, groups = o.i[14] - 256, o.i[13]
            tiles = ((B * 16 + 127) // 128) * ((o.i[6] // 16 + bnf - 1) // bnf)
            assert tiles * groups <= 256 + tiles, (B, tiles, groups)
            if not o.flags & 1:                                      # operand-type input: k_conv3_halo_sm
                assert o.i[3] % 64 == 0 and groups <= o.i[3] // 128, (B, o.i[3], groups)
        ps = [o for o in ops if o.type == U.OP_CONV and o.flags & 2]
        assert len(ps) == 3 and all(o.i[14] >= 256 and o.i[13] == 1 and not o.p[7] for o in ps), B
    net.lds_mid_min_rows = 0
    ops = U._Plan(net, 16, CPU).build().ops
    assert not any(o.type == U.OP_CONV and o.i[14] >= 256 and (o.i[13] > 1 or o.flags & 2) for o in ops)
    assert sum(o.type == U.OP_FCONV and o.i[12] == U.FNORM_GN_SELF and o.i[1] == 4 and o.i[8] == 3 for o in ops) == 16


def test_every_forced_4_slice_conv_meets_the_host_predicates_of_k_conv4_gn():
    """the planner forces S = 4 (and un-pairs the res_conv) for the 4x4 GroupNorm-self convs on ITS assumptions; the host decides
    later (csrc/fused_host.h: conv4_cs4, conv4_mb_setup) and silently falls back to the general kernel -- 4 slices + a reduce launch, correct
    but slow -- when a predicate fails.  The predicates, restated on the planned ops (sized plans: real arena offsets): 16-byte aligned
    affine operands and a float4-aligned scale/shift row stride; a lazy split-K source of at most 4 slabs; a slice of two whole groups in
    ONE source (C1 % Cs == 0) of 256 | 512 channels; B >= 2: 32-bit element offsets and, for Cs = 512, no split-K source unless B is odd."""
    from sparsefusion_amd import unet as U
    net = U.Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    net.unfused_min_rows_4 = 0                                      # keep the 4x4 level on the fused kernels at every B
    checked = 0
    for B in (1, 2, 3, 4, 6, 8, 16, 32):
        s = U._Plan(net, B, CPU).build()
        plan = U._Plan(net, B, CPU, (s.zero.off, s.misc.off + s.ws_bytes + s.ws2_bytes + 512, s.ws_bytes, s.ws2_bytes)).build()
        for k, o in enumerate(plan.ops):
            if not (o.type == U.OP_FCONV and o.i[12] == U.FNORM_GN_SELF and o.i[1] == 4 and o.i[8] == 3 and o.i[17] == 4):
                continue
            if o.flags & (16 | 32 | 64 | 128):
                continue                                            # pairs / pipelined / pooled ops are not k_conv4_gn's
            where = (B, k)
            C1, C2, S = o.i[3], o.i[4], o.i[17]
            C = C1 + C2
            Cs = C // S
            assert Cs in (256, 512) and (C // 8) * 2 == Cs, where                        # a slice = two whole groups
            assert C1 % Cs == 0, where                                                   # ... inside ONE source
            assert o.p[13] % 16 == 0 and o.p[14] % 16 == 0 and (o.p[15] or 0) % 16 == 0 and o.i[18] % 4 == 0, where      # gamma, beta, scale/shift row
            assert o.i[14:17] == [4, 1, 1][:3] or tuple(o.i[14:17]) == (4, 1, 1), where  # TR, WM, WN
            if o.i[9] == 1:
                assert o.i[10] <= 4, where                                               # lazy split-K source: <= 4 slabs
            if B >= 2 and not (o.i[19] & 1):
                assert B * 16 * max(C1, o.i[11]) < (1 << 30), where                      # 32-bit element offsets
            checked += 1
    assert checked >= 8 * 12


def test_unet_plans_of_every_batch_regime_pass_the_op_audit():
    """The op audit of test_unet_plan_invariants (operand presence, alignment contracts, split-K workspaces, lazy tensors all consumed) over the
    batch sizes at which the r06 planner rules switch kernels, ragged ones included (B = 9, 17, 33: a partly filled last tile of whole maps)."""
    from sparsefusion_amd.unet import Unet, _Plan
    net = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
               layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False)
    for B in (2, 3, 5, 8, 9, 12, 16, 17, 24, 32, 33):
        s = _Plan(net, B, CPU).build()
        plan = _Plan(net, B, CPU, (s.zero.off, s.misc.off + s.ws_bytes + s.ws2_bytes + 512, s.ws_bytes, s.ws2_bytes)).build()
        assert _audit(plan.ops, f"unet B={B}", sized=True) >= 91
        assert plan.ws_owner is None or plan.ws_owner.lazy is None
