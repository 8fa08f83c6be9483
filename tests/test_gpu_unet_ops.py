"""GPU tests of the individual UNet kernels (csrc/unet_ops.hip) through the C-ABI plan executor, each against
a plain PyTorch fp32 reference of the same op.  MFMA operands are bf16, so the references are fed the
bf16-rounded operands: what is left is fp32 accumulation order (tolerances ~1e-4 relative)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(ops):
    from sparsefusion_amd import _lib
    arr = (_lib.SfOp * len(ops))(*ops)
    _lib.check(_lib.lib().sf_plan_run(arr, len(ops), _lib.stream_ptr()), "plan")
    torch.cuda.synchronize()


_KEEP = []          # device temporaries referenced by raw pointer must outlive the launch


def _op(type_, flags=0, p=(), i=(), f=()):
    from sparsefusion_amd import _lib
    o = _lib.SfOp()
    o.type, o.flags = type_, flags
    for k, v in enumerate(p):
        if torch.is_tensor(v):
            _KEEP.append(v)
        o.p[k] = v.data_ptr() if torch.is_tensor(v) else (v or None)
    for k, v in enumerate(i):
        o.i[k] = int(v)
    for k, v in enumerate(f):
        o.f[k] = float(v)
    return o


def _pack_conv(w):
    from sparsefusion_amd import _lib
    lib = _lib.lib()
    co, ci, kh, kw = w.shape
    cpad = (ci + 31) // 32 * 32
    buf = torch.empty(lib.sf_conv_packed_elems(co, cpad, kh, kw), dtype=torch.int16)
    _lib.check(lib.sf_conv_pack_weights(w.contiguous().data_ptr(), co, ci, cpad, kh, kw, buf.data_ptr()))
    return buf.to(DEV), cpad


def bf(t):
    return t.to(torch.bfloat16).float()


CONV_CASES = [
    # B, H, Cin, Cout, k, stride, pad, groups, tile(WM,WN), a_f32, resid
    (1, 4, 1024, 1024, 3, 1, 1, 8, (1, 4), False, True),      # 4x4 weight-streaming layer, split-K
    (2, 8, 512, 512, 3, 1, 1, 2, (4, 4), False, False),
    (1, 32, 256, 256, 3, 1, 1, 2, (4, 4), False, False),
    (1, 32, 260, 64, 15, 1, 7, 16, (4, 4), True, False),      # init CrossEmbed k=15, Cin padded 260 -> 288
    (1, 32, 260, 64, 7, 1, 3, 4, (2, 2), True, False),
    (1, 16, 256, 512, 4, 2, 1, 3, (4, 2), True, False),       # Downsample conv4x4 s2 p1
    (1, 32, 256, 4, 3, 1, 1, 1, (4, 1), True, False),         # final conv: Cout 4 (N padded to 16)
    (3, 4, 768, 128, 1, 1, 0, 1, (1, 2), False, False),       # 1x1 / linear, odd batch
    (1, 4, 2048, 1024, 1, 1, 0, 5, (1, 1), False, True),
    (1, 8, 96, 48, 3, 1, 1, 1, (2, 1), False, False),         # small-config sizes: Cout 48 = 3 n-frags
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm(case):
    B, H, Cin, Cout, k, stride, pad, groups, (WM, WN), a_f32, use_res = case
    g = torch.Generator().manual_seed(Cin + Cout + k)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, H, H, generator=g)
    wp, cpad = _pack_conv(w)
    xh = torch.zeros(B, H, H, cpad)
    xh[..., :Cin] = x.permute(0, 2, 3, 1)
    xd = xh.to(DEV) if a_f32 else xh.to(torch.bfloat16).to(DEV)
    Ho = (H + 2 * pad - k) // stride + 1
    ldc, co_off = Cout + 8, 4                                                 # exercise ldc / channel offset
    out = torch.zeros(B, Ho, Ho, ldc, device=DEV)
    res = torch.randn(B, Ho, Ho, ldc, generator=g).to(DEV) if use_res else None
    ws = torch.empty(groups * B * Ho * Ho * ((Cout + 15) // 16 * 16), device=DEV) if groups > 1 else None
    _run([_op(1, 1 if a_f32 else 0, p=(xd, wp, bias.to(DEV), out, res, ws),
              i=(B, H, H, cpad, Ho, Ho, Cout, ldc, co_off, k, k, stride, pad, groups, WM * 16 + WN))])
    ref = F.conv2d(bf(x), bf(w), bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
    if use_res:
        ref = ref + res.cpu()[..., co_off:co_off + Cout]
    got = out.cpu()[..., co_off:co_off + Cout]
    assert torch.allclose(got, ref, rtol=2e-4, atol=2e-4), (got - ref).abs().max()
    assert out.cpu()[..., :co_off].abs().max() == 0 and out.cpu()[..., co_off + Cout:].abs().max() == 0


LDS_CASES = [
    # B, H, Cin, Cout, k, stride, pad, bnf, a_f32, resid, relu, ups
    (1, 64, 128, 128, 3, 1, 1, 8, False, True, False, False),     # VAE-like 3x3
    (2, 32, 64, 64, 3, 1, 1, 4, True, False, True, False),        # VGG conv1_2-like: fp32 A, ReLU epilogue, 4 n-frags
    (1, 32, 3, 64, 3, 1, 1, 4, True, False, True, False),         # Cin 3 padded to 32: KS = 9 is odd -> padded last stage
    (1, 40, 96, 200, 3, 1, 1, 8, False, False, False, False),     # ragged: M = 1600 (12.5 tiles), Cout = 200 (12.5 frags)
    (1, 64, 128, 128, 3, 2, 0, 8, True, False, False, False),     # VAE Downsample: stride 2, pad right/bottom only
    (1, 32, 128, 128, 3, 1, 1, 8, True, False, False, True),      # VAE Upsample: nearest x2 folded into the addressing
    (3, 16, 256, 384, 1, 1, 0, 8, False, True, False, False),     # 1x1, odd batch
]


@pytest.mark.parametrize("case", LDS_CASES)
def test_conv_lds_tiled(case):
    """k_conv_lds (tile code 256 + n-fragments per workgroup): same contract as k_conv_igemm without split-K."""
    B, H, Cin, Cout, k, stride, pad, bnf, a_f32, use_res, relu, ups = case
    g = torch.Generator().manual_seed(Cin + Cout + k + H)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, H, H, generator=g)
    wp, cpad = _pack_conv(w)
    xh = torch.zeros(B, H, H, cpad)
    xh[..., :Cin] = x.permute(0, 2, 3, 1)
    xd = xh.to(DEV) if a_f32 else xh.to(torch.bfloat16).to(DEV)
    Hin = 2 * H if ups else H                                         # logical input size seen by the conv
    if stride == 2:
        Ho = Hin // 2                                                 # asymmetric (0,1,0,1) padding: explicit output size
    else:
        Ho = (Hin + 2 * pad - k) // stride + 1
    ldc, co_off = Cout + 8, 4
    out = torch.zeros(B, Ho, Ho, ldc, device=DEV)
    res = torch.randn(B, Ho, Ho, ldc, generator=g).to(DEV) if use_res else None
    flags = (1 if a_f32 else 0) | (16 if ups else 0) | (32 if relu else 0)
    _run([_op(1, flags, p=(xd, wp, bias.to(DEV), out, res, None),
              i=(B, Hin, Hin, cpad, Ho, Ho, Cout, ldc, co_off, k, k, stride, pad, 1, 256 + bnf))])
    xr = bf(x)
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    if stride == 2:
        xr = F.pad(xr, (0, 1, 0, 1))
    ref = F.conv2d(xr, bf(w), bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
    if use_res:
        ref = ref + res.cpu()[..., co_off:co_off + Cout]
    if relu:
        ref = F.relu(ref)
    got = out.cpu()[..., co_off:co_off + Cout]
    assert torch.allclose(got, ref, rtol=2e-4, atol=2e-4), (got - ref).abs().max()
    assert out.cpu()[..., :co_off].abs().max() == 0 and out.cpu()[..., co_off + Cout:].abs().max() == 0
    # accumulate flag: a second launch adds onto the first result
    if not relu and not use_res:
        _run([_op(1, flags | 4, p=(xd, wp, bias.to(DEV), out, None, None),
                  i=(B, Hin, Hin, cpad, Ho, Ho, Cout, ldc, co_off, k, k, stride, pad, 1, 256 + bnf))])
        assert torch.allclose(out.cpu()[..., co_off:co_off + Cout], 2 * ref, rtol=4e-4, atol=4e-4)


GLDS_CASES = [
    # B, H, Cin, Cout, k, stride, pad, bnf, resid, relu, ups, gn
    (1, 64, 128, 128, 3, 1, 1, 8, True, False, False, True),      # SD-VAE 3x3 with GroupNorm partial sums, 32 tiles
    (1, 128, 256, 256, 3, 1, 1, 8, False, False, False, True),    # 72 stages, 256 workgroups: one per CU, two channel tiles
    (1, 64, 512, 512, 3, 1, 1, 4, False, False, False, False),    # 64-channel tiles (the 64x64 level), 144 stages
    (1, 32, 64, 64, 3, 1, 1, 4, False, True, False, False),       # 9 stages, ReLU
    (1, 40, 192, 200, 3, 1, 1, 8, False, False, False, False),    # ragged: M = 1600 (12.5 tiles), Cout = 200 (12.5 frags)
    (1, 64, 128, 128, 3, 2, 0, 8, False, False, False, False),    # Downsample addressing: stride 2, pad right/bottom only
    (1, 32, 128, 128, 3, 1, 1, 8, False, False, True, False),     # Upsample: nearest x2 folded into the addressing
    (3, 16, 256, 384, 1, 1, 0, 8, True, False, False, False),     # 1x1 (4 stages), odd batch
    (2, 16, 64, 64, 1, 1, 0, 4, False, False, False, False),      # ONE stage: shorter than the ring
]


@pytest.mark.parametrize("nst", [3, 4])
@pytest.mark.parametrize("case", GLDS_CASES)
def test_conv_glds_is_bitwise_conv_lds(case, nst):
    """k_conv_glds (csrc/conv_glds.h: LDS-DMA ring of `nst` stage buffers, loader / matrix wave specialisation, counted vmcnt) against
    k_conv_lds on the same operands, both selected explicitly by tile code (256 + 16 * selector + n-fragments): bit-identical output,
    GroupNorm partial sums to summation order, repeated launches included (a race in the ring would show as run-to-run differences)."""
    B, H, Cin, Cout, k, stride, pad, bnf, use_res, relu, ups, gn = case
    g = torch.Generator().manual_seed(Cin + Cout + k + H + nst)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, H, H, generator=g)
    wp, cpad = _pack_conv(w)
    assert cpad == Cin
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    Hin = 2 * H if ups else H
    Ho = Hin // 2 if stride == 2 else (Hin + 2 * pad - k) // stride + 1
    M = B * Ho * Ho
    ldc, co_off = (Cout, 0) if gn else (Cout + 8, 4)
    res = torch.randn(B, Ho, Ho, ldc, generator=g).to(DEV) if use_res else None
    cg = 4 if Cout <= 128 else 8
    flags = (16 if ups else 0) | (32 if relu else 0) | (128 if gn else 0)
    outs = []
    for sel in (1, nst, nst, nst):
        out = torch.zeros(B, Ho, Ho, ldc, device=DEV)
        part = torch.full((max(M // 128, 1), Cout // cg, 2), float("nan"), dtype=torch.float64, device=DEV) if gn else None
        _run([_op(1, flags, p=(xd, wp, bias.to(DEV), out, res, None, part),
                  i=(B, Hin, Hin, cpad, Ho, Ho, Cout, ldc, co_off, k, k, stride, pad, 1, 256 + 16 * sel + bnf, cg if gn else 0))])
        torch.cuda.synchronize()
        outs.append((out.cpu(), part.cpu() if gn else None))
    ref, rpart = outs[0]
    assert ref.abs().max() > 0
    for got, gpart in outs[1:]:
        assert torch.equal(got, ref), float((got - ref).abs().max())
        if gn:                                                        # another (fixed) summation order than k_conv_lds_gn; identical run to run
            assert torch.allclose(gpart, rpart, rtol=1e-5, atol=1e-4) and torch.equal(gpart, outs[1][1])
    # ... and both against torch itself (kernel-vs-kernel alone would pass a bug the two share on a shape LDS_CASES lacks)
    xr = bf(x)
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    if stride == 2:
        xr = F.pad(xr, (0, 1, 0, 1))
    want = F.conv2d(xr, bf(w), bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
    if use_res:
        want = want + res.cpu()[..., co_off:co_off + Cout]
    if relu:
        want = F.relu(want)
    got = outs[1][0]
    assert torch.allclose(got[..., co_off:co_off + Cout], want, rtol=2e-4, atol=2e-4), float((got[..., co_off:co_off + Cout] - want).abs().max())
    if co_off:
        assert got[..., :co_off].abs().max() == 0 and got[..., co_off + Cout:].abs().max() == 0
    if gn:                                                            # per-image, per-group (sum, sum of squares) of the written output
        tpi = max(Ho * Ho // 128, 1)
        v = got.double().view(B, Ho * Ho, Cout // cg, cg)
        direct = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)
        assert torch.allclose(outs[1][1].view(B, tpi, Cout // cg, 2).sum(1), direct, rtol=1e-6, atol=1e-3)


HALO_CASES = [
    # B, H, W, Cin, Cout, bnf, resid, relu, gn (+ optional: ups)
    (1, 64, 64, 128, 128, 8, True, False, True),       # SD-VAE 3x3 with GroupNorm partial sums: 32 tiles of 8 x 16 pixels
    (1, 128, 128, 256, 256, 8, False, False, True),    # 4 chunks x 9 taps, 256 workgroups, two channel tiles
    (1, 64, 64, 512, 512, 4, False, False, False),     # 64-channel tiles, 8 chunks
    (2, 32, 48, 192, 200, 8, True, True, False),       # two images, W = 48 (3 tile columns), ragged Cout, residual + ReLU
    (1, 8, 16, 64, 64, 4, False, False, False),        # ONE tile: every halo edge lies outside the image
    (1, 64, 64, 256, 128, 8, False, False, True, 1),   # Upsample: stored 32 x 32, the halo tile of its nearest-x2 view; also fills the twin
]


@pytest.mark.parametrize("sel", [6, 7])
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3_halo_matches_conv_lds(case, sel):
    """k_conv3_halo (csrc/conv_halo.h: halo tile of a 64-channel chunk staged once, nine shifted window reads, chunk-major K) against
    k_conv_lds on the same operands: equal to fp32 reassociation, per-image GroupNorm sums equal, identical run to run."""
    B, H, W, Cin, Cout, bnf, use_res, relu, gn = case[:9]
    ups = case[9] if len(case) > 9 else 0
    g = torch.Generator().manual_seed(Cin + Cout + H + sel)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, H >> ups, W >> ups, generator=g)
    wp, cpad = _pack_conv(w)
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    M = B * H * W
    ldc, co_off = (Cout, 0) if gn else (Cout + 8, 4)
    res = torch.randn(B, H, W, ldc, generator=g).to(DEV) if use_res else None
    cg = 4 if Cout <= 128 else 8
    flags = (16 if ups else 0) | (32 if relu else 0) | (128 if gn else 0)
    outs, twins = [], []
    for s_ in (1, sel, sel, sel):
        out = torch.zeros(B, H, W, ldc, device=DEV)
        part = torch.full((M // 128, Cout // cg, 2), float("nan"), dtype=torch.float64, device=DEV) if gn else None
        twin = torch.zeros(M, Cout, dtype=torch.bfloat16, device=DEV) if ups else None          # (the twin epilogue rides on the ups case)
        twins.append(twin)
        _run([_op(1, flags, p=(xd, wp, bias.to(DEV), out, res, twin, part),
                  i=(B, H, W, cpad, H, W, Cout, ldc, co_off, 3, 3, 1, 1, 1, 256 + 16 * s_ + bnf, cg if gn else 0))])
        torch.cuda.synchronize()
        outs.append((out.cpu(), part.cpu() if gn else None))
    ref, rpart = outs[0]
    assert ref.abs().max() > 0
    for got, gpart in outs[1:]:
        assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5), float((got - ref).abs().max())
        assert torch.equal(got, outs[1][0])
        if gn:                                                        # other pixel tiles: compare the per-image sums
            tpi = H * W // 128
            assert torch.allclose(gpart.view(B, tpi, -1, 2).sum(1), rpart.view(B, tpi, -1, 2).sum(1), rtol=1e-6, atol=1e-3)
            assert torch.equal(gpart, outs[1][1])
    xr, wr = bf(x), bf(w)
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xr, wr, bias, padding=1).permute(0, 2, 3, 1)
    if use_res:
        want = want + res.cpu()[..., co_off:co_off + Cout]
    if relu:
        want = F.relu(want)
    assert torch.allclose(outs[1][0][..., co_off:co_off + Cout], want, rtol=2e-4, atol=2e-4)
    if ups:                                                           # the operand-type twin = the written output, rounded; both kernels
        for tw, (o_, _) in zip(twins, outs):
            assert torch.equal(tw.cpu().view(B, H, W, Cout), o_.to(torch.bfloat16))


LDS_SPLITK_CASES = [
    # B, H, Cin, Cout, k, pad, bnf, sel (0: default = k_conv3_halo_sm for a 3x3 on whole 4x4 / 8x8 maps, else k_conv_glds when it can; 1: k_conv_lds;
    #                                    3 / 4: k_conv_glds ring depth; 8 / 9: k_conv3_halo_sm ring depth), a_f32, groups, resid, deferred
    (8, 4, 1024, 1024, 3, 1, 4, 4, False, 16, False, False),    # the 4x4 level at B = 8: ONE pixel tile, 16 channel tiles, 16 groups of 9 stages
    (8, 4, 1024, 1024, 3, 1, 4, 0, False, 8, False, False),     # ... on k_conv3_halo_sm: 8 whole maps per tile, 8 groups of two 64-channel chunks
    (32, 4, 1024, 1024, 3, 1, 4, 3, False, 4, True, False),     # B = 32: 4 x 16 tiles, 4 groups of 36 stages; residual in the reduction
    (32, 4, 1024, 1024, 3, 1, 4, 9, False, 4, True, False),     # ... k_conv3_halo_sm, 4 chunks per group
    (32, 4, 2048, 1024, 3, 1, 8, 4, False, 8, False, True),     # conv1 of an up block on the concat; reduction deferred: the slabs are checked
    (32, 4, 2048, 1024, 3, 1, 8, 8, False, 8, False, True),     # ... k_conv3_halo_sm, 3-deep ring
    (11, 4, 1024, 200, 3, 1, 8, 0, False, 3, False, False),     # ragged: 11 maps (the second tile holds 3), Cout = 200, 16 chunks in 3 groups (5 / 5 / 6)
    (32, 8, 1024, 1024, 3, 1, 8, 0, False, 2, False, False),    # the 8x8 level at B = 32: 2 whole maps per tile, 16 x 8 tiles, 2 groups
    (5, 8, 512, 512, 3, 1, 4, 9, False, 1, True, False),        # 8x8, no split-K: the epilogue writes the output (bias, residual); 5 maps = 2.5 tiles
    (9, 4, 1024, 640, 1, 0, 4, 0, False, 3, False, False),      # 1x1, ragged M = 144, 16 stages in 3 groups (5 / 5 / 6)
    (8, 4, 512, 1024, 3, 1, 8, 1, True, 4, False, False),       # k_conv_lds, fp32 activations
    (16, 4, 96, 200, 3, 1, 8, 1, False, 3, True, False),        # KS = 27 is odd (padded last stage), ragged Cout
]


@pytest.mark.parametrize("case", LDS_SPLITK_CASES)
def test_conv_lds_tiled_split_k(case):
    """r06: split-K groups on the LDS-tiled kernels (the 4x4 level of the B >= 8 plans; k_conv3_halo_sm on whole 4x4 / 8x8 maps, k_conv_glds,
    k_conv_lds): workspace [group][row][npad] as k_conv_igemm leaves it, reduced by k_splitk_reduce (bias, residual) or left to the consumer
    (flag 8) -- against conv2d on the same rounded operands."""
    B, H, Cin, Cout, k, pad, bnf, sel, a_f32, groups, use_res, deferred = case
    g = torch.Generator().manual_seed(Cin + Cout + k + B)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, H, H, generator=g)
    wp, cpad = _pack_conv(w)
    assert cpad == Cin
    xh = x.permute(0, 2, 3, 1).contiguous()
    xd = xh.to(DEV) if a_f32 else xh.to(torch.bfloat16).to(DEV)
    M, npad = B * H * H, (Cout + 15) // 16 * 16
    out = torch.full((M, Cout), float("nan"), device=DEV)
    res = torch.randn(M, Cout, generator=g).to(DEV) if use_res else None
    ws = torch.full((groups, M, npad), float("nan"), device=DEV) if groups > 1 else None
    _run([_op(1, (1 if a_f32 else 0) | (8 if deferred else 0), p=(xd, wp, bias.to(DEV), out, res, ws),
              i=(B, H, H, Cin, H, H, Cout, Cout, 0, k, k, 1, pad, groups, 256 + 16 * sel + bnf))])
    ref = F.conv2d(bf(x), bf(w), None, padding=pad).permute(0, 2, 3, 1).reshape(M, Cout)
    if groups == 1:
        ref = ref + bias + (res.cpu() if use_res else 0)
        assert torch.allclose(out.cpu(), ref, rtol=2e-4, atol=3e-4), (out.cpu() - ref).abs().max()
        return
    assert not bool(torch.isnan(ws[:, :, :Cout]).any())
    assert torch.allclose(ws.cpu()[:, :, :Cout].sum(0), ref, rtol=2e-4, atol=3e-4), (ws.cpu()[:, :, :Cout].sum(0) - ref).abs().max()
    if deferred:
        assert bool(torch.isnan(out).all())
        return
    ref = ref + bias
    if use_res:
        ref = ref + res.cpu()
    assert torch.allclose(out.cpu(), ref, rtol=2e-4, atol=3e-4), (out.cpu() - ref).abs().max()


@pytest.mark.parametrize("B,H,Cin,Cout,bnf,sel,a_f32", [
    (32, 4, 1024, 4096, 8, 1, True),        # ups.0: 4x4 -> 8x8 (k_conv_lds: the UNet hands the Upsample its fp32 residual stream)
    (8, 8, 1024, 2048, 8, 1, True),         # ups.1 at B = 8
    (8, 16, 512, 1024, 8, 0, False),        # k_conv_glds (operand-type activations)
    (3, 4, 128, 72, 4, 0, False),           # ragged M = 48, Cout = 72 (18 output channels)
])
def test_conv_lds_tiled_pixel_shuffle(B, H, Cin, Cout, bnf, sel, a_f32):
    """r06: the SiLU + PixelShuffle(2) epilogue of an Upsample (imagen_pytorch.py:578-606) on the LDS-tiled kernels, into a channel
    window of a wider tensor (ldc / co_off), against torch pixel_shuffle."""
    g = torch.Generator().manual_seed(Cin + Cout + B)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, H, H, generator=g)
    wp, _ = _pack_conv(w)
    xh = x.permute(0, 2, 3, 1).contiguous()
    xd = xh.to(DEV) if a_f32 else xh.to(torch.bfloat16).to(DEV)
    ldc, co_off = Cout // 4 + 8, 4
    up = torch.full((B, 2 * H, 2 * H, ldc), float("nan"), device=DEV)
    _run([_op(1, (1 if a_f32 else 0) | 2, p=(xd, wp, bias.to(DEV), up, None, None),
              i=(B, H, H, Cin, H, H, Cout, ldc, co_off, 1, 1, 1, 0, 1, 256 + 16 * sel + bnf))])
    ref = F.pixel_shuffle(F.silu(F.conv2d(bf(x), bf(w), bias)), 2).permute(0, 2, 3, 1)
    got = up.cpu()
    assert torch.allclose(got[..., co_off:co_off + Cout // 4], ref, rtol=2e-4, atol=3e-4), (got[..., co_off:co_off + Cout // 4] - ref).abs().max()
    assert bool(torch.isnan(got[..., :co_off]).all()) and bool(torch.isnan(got[..., co_off + Cout // 4:]).all())


def test_conv_accumulates_and_pixel_shuffle():
    g = torch.Generator().manual_seed(3)
    B, H, C = 2, 8, 128
    x = torch.randn(B, C, H, H, generator=g)
    w3, w1 = torch.randn(C, C, 3, 3, generator=g) / 34, torch.randn(C, C, 1, 1, generator=g) / 11
    b3, b1 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    p3, _ = _pack_conv(w3)
    p1, _ = _pack_conv(w1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.full((B, H, H, C), float("nan"), device=DEV)      # the first conv stores, the second accumulates (flag 4)
    ws = torch.empty(2 * B * H * H * C, device=DEV)
    common = (B, H, H, C, H, H, C, C, 0)
    _run([_op(1, 1, p=(xd, p3, b3.to(DEV), out, None, ws), i=common + (3, 3, 1, 1, 2, 2 * 16 + 4)),
          _op(1, 1 | 4, p=(xd, p1, b1.to(DEV), out, None, None), i=common + (1, 1, 1, 0, 1, 2 * 16 + 4))])   # Parallel(conv3x3, conv1x1)
    ref = (F.conv2d(bf(x), bf(w3), b3, padding=1) + F.conv2d(bf(x), bf(w1), b1)).permute(0, 2, 3, 1)
    assert torch.allclose(out.cpu(), ref, rtol=2e-4, atol=2e-4)
    # PixelShuffleUpsample: conv1x1 -> SiLU -> PixelShuffle(2)
    wu, bu = torch.randn(4 * 64, C, 1, 1, generator=g) / 11, torch.randn(4 * 64, generator=g)
    pu, _ = _pack_conv(wu)
    up = torch.full((B, 2 * H, 2 * H, 64), float("nan"), device=DEV)
    _run([_op(1, 1 | 2, p=(xd, pu, bu.to(DEV), up, None), i=(B, H, H, C, H, H, 256, 64, 0, 1, 1, 1, 0, 1, 2 * 16 + 2))])
    ref = F.pixel_shuffle(F.silu(F.conv2d(bf(x), bf(wu), bu)), 2).permute(0, 2, 3, 1)
    assert torch.allclose(up.cpu(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("B,H,C1,C2,with_ss", [(1, 32, 256, 256, True), (2, 8, 1024, 512, False), (1, 4, 1024, 0, True),
                                                 (3, 16, 64, 0, True), (1, 32, 512, 0, False),
                                                 # r06: k_gn_one (one launch; <256 | 1024 threads, 2 .. 16 float4 per thread>; the plans take it from B * 8 >= 256 workgroups on, flag 8 forces it)
                                                 (32, 8, 1024, 0, True),
                                                 (8, 4, 1024, 0, True), (9, 4, 1024, 1024, False), (8, 8, 1024, 512, True), (8, 16, 512, 256, True),
                                                 (8, 32, 256, 0, True), (8, 32, 256, 256, False), (16, 16, 512, 0, True)])
def test_gn_act(B, H, C1, C2, with_ss):
    g = torch.Generator().manual_seed(C1 + C2 + H)
    HW, C = H * H, C1 + C2
    x1 = torch.randn(B, HW, C1, generator=g) * 2 + 0.5
    x2 = torch.randn(B, HW, C2, generator=g) if C2 else None
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    ss_all = torch.randn(B, 3 * C, generator=g) * 0.3
    out = torch.empty(B, HW, C, dtype=torch.bfloat16, device=DEV)
    raw = torch.empty(B, HW, C, dtype=torch.bfloat16, device=DEV)
    ssd = ss_all.to(DEV)
    ss_ptr = ssd.data_ptr() + C * 4 if with_ss else 0        # the block's slice starts at column C
    stats = torch.zeros(B * 8 * 2, dtype=torch.float64, device=DEV)
    _run([_op(2, 8 if B >= 8 else 0, p=(x1.to(DEV), x2.to(DEV) if C2 else None, gamma.to(DEV), beta.to(DEV), ss_ptr, out, raw, stats),      # flag 8: k_gn_one below its 256-workgroup rule
              i=(B, HW, C1, C2, 3 * C), f=(1e-5, 2 ** -0.5))])
    xc = torch.cat([x1, x2 * 2 ** -0.5], -1) if C2 else x1
    ref = F.group_norm(xc.permute(0, 2, 1).reshape(B, C, H, H), 8, gamma, beta, eps=1e-5)
    if with_ss:
        sc, sh = ss_all[:, C:2 * C], ss_all[:, 2 * C:3 * C]
        ref = ref * (sc[:, :, None, None] + 1) + sh[:, :, None, None]
    ref = F.silu(ref).reshape(B, C, HW).permute(0, 2, 1)
    assert torch.allclose(out.float().cpu(), ref, rtol=1.2e-2, atol=1e-2)       # bf16 output rounding (2^-8)
    assert (out.float().cpu() - ref).abs().mean() < 2e-3
    assert torch.equal(raw.cpu(), xc.to(torch.bfloat16))


@pytest.mark.parametrize("R,C,gelu,bias,f32,res", [(16, 1024, False, False, False, False), (32, 2048, True, False, False, False),
                                                     (4, 256, False, True, True, False), (48, 1024, False, False, True, True),
                                                     (7, 64, False, False, True, False), (16, 1024, 4, True, True, True), (300, 512, False, False, True, False),
                                                     # ((16 | 32 | 48) x (1024 | 2048): k_layernorm_wave since r05; gelu = 4 = op flag 4: k_layernorm on that shape; 300 rows: k_layernorm)
                                                     (4099, 256, False, True, True, True), (2048, 256, True, False, False, False)])      # k_layernorm_w256 (EFT: many 256-channel rows)
def test_layernorm(R, C, gelu, bias, f32, res):
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g) * 3 + 1
    gain, b = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    r = torch.randn(R, C, generator=g)
    out = torch.empty(R, C, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    _run([_op(3, int(gelu) | (2 if f32 else 0), p=(x.to(DEV), gain.to(DEV), b.to(DEV) if bias else None, out,
                                                 r.to(DEV) if res else None), i=(R, C), f=(1e-5,))])
    y = F.gelu(x) if (int(gelu) & 1) else x
    ref = (y - y.mean(-1, keepdim=True)) * (y.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * gain
    if bias:
        ref = ref + b
    if res:
        ref = ref + r
    if f32:
        assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=2e-5)
    else:
        assert torch.allclose(out.float().cpu(), ref, rtol=1.2e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K,in_silu,act", [(1, 1024, 17, False, 1), (2, 5000, 1024, True, 0), (8, 128, 256, False, 0),
                                                (4, 512, 1024, False, 2), (1, 33, 40, False, 0)])
def test_gemv(M, N, K, in_silu, act):
    g = torch.Generator().manual_seed(N + K)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    x = torch.randn(M, K + 3, generator=g)
    Kp = (K + 7) // 8 * 8
    wp = F.pad(w, (0, Kp - K)).to(torch.bfloat16).contiguous().to(DEV)
    y = torch.zeros(M, N + 2, device=DEV)
    _run([_op(4, (1 if in_silu else 0) | (act << 1) | 8, p=(x.to(DEV), wp, b.to(DEV), y), i=(M, N, K, Kp, K + 3, N + 2))])      # flag 8: k_gemv also at 8 rows (r06: k_gemm_rows_ks takes those)
    xin = x[:, :K]
    ref = F.linear(F.silu(xin) if in_silu else xin, bf(w), b)
    ref = F.silu(ref) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
    assert torch.allclose(y.cpu()[:, :N], ref, rtol=1e-4, atol=1e-4)
    # rows are independent: every row evaluated alone (M = 1 launch) gives the same bits as inside the M-row launch
    for m in range(M):
        y1 = torch.zeros(1, N + 2, device=DEV)
        _run([_op(4, (1 if in_silu else 0) | (act << 1), p=(x[m:m + 1].contiguous().to(DEV), wp, b.to(DEV), y1), i=(1, N, K, Kp, K + 3, N + 2))])
        assert torch.equal(y1.cpu()[0, :N], y.cpu()[m, :N]), m


@pytest.mark.parametrize("first_form", [False, True])
@pytest.mark.parametrize("M,N,K,in_silu,act", [(51, 1024, 17, False, 1), (51, 4160, 1024, True, 0), (20, 136, 300, False, 2),
                                               (32, 512, 1024, False, 1), (32, 1024, 512, False, 2), (9, 256, 125, True, 0),
                                               (8, 512, 1024, False, 1), (8, 1024, 512, False, 2)])      # 8 rows: the GlobalContext MLPs of a B = 8 hybrid block
def test_gemv_many_rows_on_mfma(M, N, K, in_silu, act, first_form):
    """OP_GEMV with 9..64 rows (a sampler's time table, Unet.time_table; the GlobalContext MLPs of a large batch) runs on k_gemm_rows /
    k_gemm_rows_ks (csrc/gemm_rows.h; N <= 4096: the K-sliced form, flag 8 forces the first one): rows on the MFMA M side, x split into
    bf16 hi + lo.  Same arithmetic as k_gemv (fp32 x times bf16 w) to ~1e-5, row by row.  (float4 staging where the row stride is a
    multiple of 4 and K of 8.)"""
    g = torch.Generator().manual_seed(N + K)
    ldx = K if K % 8 == 0 and N <= 4096 else K + 3
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    x = torch.randn(M, ldx, generator=g)
    Kp = (K + 7) // 8 * 8
    wp = F.pad(w, (0, Kp - K)).to(torch.bfloat16).contiguous().to(DEV)
    y = torch.full((M, N + 2), float("nan"), device=DEV)
    flags = (1 if in_silu else 0) | (act << 1) | (8 if first_form else 0)
    _run([_op(4, flags, p=(x.to(DEV), wp, b.to(DEV), y), i=(M, N, K, Kp, ldx, N + 2))])
    xin = x[:, :K]
    ref = F.linear(F.silu(xin) if in_silu else xin, bf(w), b)
    ref = F.silu(ref) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
    assert torch.allclose(y.cpu()[:, :N], ref, rtol=1e-4, atol=1e-4), float((y.cpu()[:, :N] - ref).abs().max())
    assert bool(torch.isnan(y[:, N:]).all())
    y8 = torch.zeros(8, N + 2, device=DEV)                    # the 8-row kernel on the first rows: the two kernels agree
    _run([_op(4, flags, p=(x[:8].contiguous().to(DEV), wp, b.to(DEV), y8), i=(8, N, K, Kp, ldx, N + 2))])
    assert torch.allclose(y8.cpu()[:, :N], y.cpu()[:8, :N], rtol=3e-5, atol=3e-5)


def test_attention_core_self_and_cross():
    g = torch.Generator().manual_seed(9)
    B, heads, dh = 2, 8, 64
    q = torch.randn(B * 16, 512, generator=g)
    kv = torch.randn(B * 16, 128, generator=g)
    ckv = torch.randn(B * 2, 128, generator=g)
    null = torch.randn(2, 64, generator=g)
    out = torch.empty(B * 16, 512, dtype=torch.bfloat16, device=DEV)
    qd, kvd, ckvd, nd = q.to(DEV), kv.to(DEV), ckv.to(DEV), null.to(DEV)
    _run([_op(5, 0, p=(qd, out, ckvd, ckvd.data_ptr() + 256, nd, nd.data_ptr() + 256, kvd, kvd.data_ptr() + 256),
              i=(B, heads, 512, 0, 2, 128, 256, 0, 1, 0, 0, 0, 16, 128, 2048, 0), f=(dh ** -0.5,))])
    qh = q.view(B, 16, heads, dh).permute(0, 2, 1, 3) * dh ** -0.5
    k = torch.cat([ckv[:, :64].view(B, 2, 64), null[0].view(1, 1, 64).expand(B, 1, 64), kv[:, :64].view(B, 16, 64)], 1)
    v = torch.cat([ckv[:, 64:].view(B, 2, 64), null[1].view(1, 1, 64).expand(B, 1, 64), kv[:, 64:].view(B, 16, 64)], 1)
    att = torch.einsum('bhid,bjd->bhij', qh, k).softmax(-1)
    ref = torch.einsum('bhij,bjd->bhid', att, v).permute(0, 2, 1, 3).reshape(B * 16, 512)
    assert torch.allclose(out.float().cpu(), ref, rtol=1.2e-2, atol=1e-2)
    # cross attention: per-head k/v from 2 context tokens + shared null
    kvc = torch.randn(B * 2, 1024, generator=g).to(DEV)
    _run([_op(5, 0, p=(qd, out, nd, nd.data_ptr() + 256, kvc, kvc.data_ptr() + 2048, None, None),
              i=(B, heads, 512, 0, 1, 0, 0, 0, 2, 1024, 2048, 64, 0, 0, 0, 0), f=(dh ** -0.5,))])
    kc = kvc.cpu()[:, :512].view(B, 2, heads, dh).permute(0, 2, 1, 3)
    vc = kvc.cpu()[:, 512:].view(B, 2, heads, dh).permute(0, 2, 1, 3)
    kk = torch.cat([null[0].view(1, 1, 1, 64).expand(B, heads, 1, 64), kc], 2)
    vv = torch.cat([null[1].view(1, 1, 1, 64).expand(B, heads, 1, 64), vc], 2)
    att = torch.einsum('bhid,bhjd->bhij', qh, kk).softmax(-1)
    ref = torch.einsum('bhij,bhjd->bhid', att, vv).permute(0, 2, 1, 3).reshape(B * 16, 512)
    assert torch.allclose(out.float().cpu(), ref, rtol=1.2e-2, atol=1e-2)


def test_gca_pool_gate_and_layout_ops():
    g = torch.Generator().manual_seed(10)
    B, HW, C = 2, 256, 512
    h = torch.randn(B, HW, C, generator=g)
    wk, bk = torch.randn(C, generator=g) / 8, torch.randn(1, generator=g)
    pooled = torch.zeros(B, C, device=DEV)          # accumulated with atomics: the caller zeroes it
    hd = h.to(DEV)
    _run([_op(6, 0, p=(hd, wk.to(DEV), bk.to(DEV), pooled, torch.empty(B * HW, device=DEV)), i=(B, HW, C))])
    att = (h @ wk + bk).softmax(-1)
    assert torch.allclose(pooled.cpu(), torch.einsum('bn,bnc->bc', att, h), rtol=1e-4, atol=1e-5)
    gate, res = torch.rand(B, C, generator=g), torch.randn(B, HW, C, generator=g)
    out = torch.empty(B, HW, C, device=DEV)
    _run([_op(7, 1, p=(hd, gate.to(DEV), res.to(DEV), out), i=(B, HW, C))])
    assert torch.allclose(out.cpu(), h * gate[:, None] + res, rtol=1e-6, atol=1e-6)
    out2 = res.to(DEV).clone()
    _run([_op(7, 1, p=(hd, gate.to(DEV), None, out2), i=(B, HW, C))])      # residual already in `out`
    assert torch.allclose(out2.cpu(), h * gate[:, None] + res, rtol=1e-6, atol=1e-6)
    cond, x = torch.randn(B, 60, 1024, generator=g), torch.randn(B, 4, 1024, generator=g)
    packed = torch.full((B, 1024, 64), float("nan"), device=DEV)
    _run([_op(7, 2, p=(cond.to(DEV), x.to(DEV), None, packed), i=(B, 1024, 60, 4, 64))])
    assert torch.equal(packed.cpu(), torch.cat([cond, x], 1).permute(0, 2, 1))
    nhwc = torch.randn(B, 1024, 4, generator=g)
    un = torch.empty(B, 4, 1024, device=DEV)
    _run([_op(7, 3, p=(nhwc.to(DEV), None, None, un), i=(B, 1024, 4, 4))])
    assert torch.equal(un.cpu(), nhwc.permute(0, 2, 1))
    t, w = torch.tensor([0.3, -2.5]), torch.randn(8, generator=g)
    emb = torch.empty(2, 17, device=DEV)
    _run([_op(9, 0, p=(t.to(DEV), w.to(DEV), None, emb), i=(2, 8))])
    fr = t[:, None] * w[None] * 2 * torch.pi
    assert torch.allclose(emb.cpu(), torch.cat([t[:, None], fr.sin(), fr.cos()], -1), atol=2e-6)


def test_plms_update_kernels():
    from sparsefusion_amd import _lib
    from sparsefusion_amd.plms import step_coefficients
    from oracle import unet_ref
    g = torch.Generator().manual_seed(12)
    x, e, nz = (torch.randn(2, 4, 32, 32, generator=g) for _ in range(3))
    for t, tn in ((0.5, 0.49), (0.02, 0.0), (0.97, 0.5)):
        coef = step_coefficients(t, tn, 10.0)
        xp, x0 = torch.empty_like(x, device=DEV), torch.empty_like(x, device=DEV)
        xd, ed, nd = x.to(DEV), e.to(DEV), nz.to(DEV)
        _lib.check(_lib.lib().sf_plms_update(_lib.ptr(xd), _lib.ptr(ed), _lib.ptr(nd), coef.ctypes.data, x.numel(), _lib.ptr(xp),
                                             _lib.ptr(x0), _lib.stream_ptr()))
        torch.cuda.synchronize()
        tb, tnb = torch.full((2, 1, 1, 1), t), torch.full((2, 1, 1, 1), tn)
        a, s = unet_ref.alpha_sigma(unet_ref.log_snr(tb))
        xs = ((x - s * e) / a.clamp(min=1e-8)).clamp(-10, 10)
        mean, _, logvar = unet_ref.q_posterior(xs, x, tb, tnb)
        ref = mean + (0.0 if tn == 0 else 1.0) * (0.5 * logvar).exp() * nz
        assert torch.allclose(x0.cpu(), xs, rtol=1e-6, atol=1e-6) and torch.allclose(xp.cpu(), ref, rtol=1e-5, atol=1e-6)
        # in place (x_prev aliases x), as plms.py runs it on the plan's input buffer: same values bit for bit
        xa, x0a = xd.clone(), torch.empty_like(x0)
        _lib.check(_lib.lib().sf_plms_update(_lib.ptr(xa), _lib.ptr(ed), _lib.ptr(nd), coef.ctypes.data, x.numel(), _lib.ptr(xa),
                                             _lib.ptr(x0a), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(xa, xp) and torch.equal(x0a, x0)


@pytest.mark.parametrize("B,R,Cx,cws", [(1, 32, 4, (128, 64, 64)), (2, 16, 3, (64, 32, 32))])
def test_init_x_direct_conv(B, R, Cx, cws):
    """SF_OP_INITX (csrc/initx.hip): x0 = base + CrossEmbed(x) for the latent channels, three convs k = 3 / 7 / 15 into channel
    slices (external/imagen_pytorch.py:1017-1042): MFMA on bf16 operands vs torch conv2d on the same bf16-rounded operands."""
    from sparsefusion_amd.unet import init_x_weight_table
    g = torch.Generator().manual_seed(11)
    ks = (3, 7, 15)
    dim = sum(cws)
    x = torch.randn(B, Cx, R, R, generator=g)
    ws = [torch.randn(cw, Cx, k, k, generator=g) / (Cx * k * k) ** 0.5 for cw, k in zip(cws, ks)]
    base = torch.randn(B * R * R, dim, generator=g)
    want = torch.cat([F.conv2d(bf(x), bf(w), padding=k // 2) for w, k in zip(ws, ks)], 1).permute(0, 2, 3, 1).reshape(B * R * R, dim) + base
    wt, woffs = init_x_weight_table(ws)
    offs = [0, cws[0], cws[0] + cws[1]]
    xd, bd, wd = x.to(DEV), base.to(DEV), wt.to(DEV)
    out = torch.full((B * R * R, dim), float("nan"), device=DEV)
    _run([_op(17, 0, p=(xd, bd, wd, out), i=(B, R, R, Cx, dim) + tuple(cws) + tuple(offs) + tuple(woffs))])
    assert torch.allclose(out.cpu(), want, rtol=1e-4, atol=2e-4), float((out.cpu() - want).abs().max())
