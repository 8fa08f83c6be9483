"""Shared cases of the fused UNet kernels (sparsefusion_amd/csrc/fused_kernels.h): the same op descriptors run either on
CPU threads (backend "emu": tests/hostemu, kernel logic without a GPU) or through the C ABI on the GPU (backend "gpu":
sf_plan_run), and are compared with a plain torch fp32 reference of the same op on bf16-rounded operands
(GroupNorm / LayerNorm -> scale/shift -> SiLU -> conv, external/imagen_pytorch.py:641-662)."""
import torch
import torch.nn.functional as F

from hostemu import fused
from sparsefusion_amd import _lib

OP_FCONV, OP_SLOTS, OP_GCA = 14, 15, 16
NONE, GN_SELF, GN_SLOTS, LN, ATTN = 0, 1, 2, 3, 4


def bf(x):
    return x.to(torch.bfloat16).float()


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-20))


def slots_of(x, M, C):
    """[M/16][C/16][2] (sum, sum of squares) over 16-pixel x 16-channel blocks of an NHWC [M, C] tensor."""
    t = x.reshape(M // 16, 16, C // 16, 16).permute(0, 2, 1, 3).reshape(M // 16, C // 16, 256)
    return torch.stack([t.sum(-1), (t * t).sum(-1)], -1).contiguous()


def nhwc(x, B, H, W):      # [M, C] -> NCHW
    return x.reshape(B, H, W, -1).permute(0, 3, 1, 2)


def to_rows(x):            # NCHW -> [M, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


TIMING_LIB = None          # tools/fconv_phases.py: the instrumented build of unet_fused.hip (libsf_fused_timing.so)


def run_ops(ops, backend):
    if backend == "emu":
        fused.run(ops)
    elif TIMING_LIB is not None:
        for o in ops:
            if TIMING_LIB.sf_fused_op_run(_lib.C.byref(o), _lib.stream_ptr()):
                raise RuntimeError("timing lib: op failed")
        torch.cuda.synchronize()
    else:
        arr = (_lib.SfOp * len(ops))(*ops)
        _lib.check(_lib.lib().sf_plan_run(arr, len(ops), _lib.stream_ptr()), "plan")
        torch.cuda.synchronize()


def run_conv_case(backend, B, H, W, C1, C2, Cout, k, norm, WM, WN, S=1, lazy=0, silu=True, ss=True, accum=False, resid=False,
                  slots=True, pre_gelu=False, ln_bias=False, seed=0, G=8, scale2=2 ** -0.5, tol=4e-3, dbg=None, reps=1, logits=False,
                  out_gelu=False, pair=False, pipe=False, pool=False, general=False, one_image=False, tw=None, keep_pipe=False):
    """tw (r06): run the op on k_conv3s with a tw-pixel-wide tile (W = full-width strips): plain output rows, op field i[19] = tw << 2;
    keep_pipe: the same op with i[19] bit 1 set = the general pipelined kernel."""
    dev = "cpu" if backend == "emu" else "cuda:0"
    d = lambda t: None if t is None else t.to(dev)
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    M, HW, C = B * H * W, H * W, C1 + C2
    TR = 16 * WM // W
    # ---- source 1 (possibly lazy) and source 2
    if lazy == 1:
        groups, npad = 3, C1 + 16
        ws = rn(groups, M, npad) * 0.5
        bias1, r1 = rn(C1), rn(M, C1)
        x1 = ws[:, :, :C1].sum(0) + bias1 + r1
        s1 = dict(p=d(torch.full((M, C1), float("nan"))), a=d(ws), b=d(bias1), r=d(r1), mode=1, groups=groups, npad=npad)
    elif lazy == 2:
        hh, gate, r1 = rn(M, C1), torch.sigmoid(rn(B, C1)), rn(M, C1)
        x1 = hh * gate.repeat_interleave(HW, 0) + r1
        s1 = dict(p=d(torch.full((M, C1), float("nan"))), a=d(hh), b=d(gate), r=d(r1), mode=2, groups=0, npad=0)
    else:
        x1 = rn(M, C1) * 1.5 + 0.3
        s1 = dict(p=d(x1.clone()), a=None, b=None, r=None, mode=0, groups=0, npad=0)
    x2 = rn(M, C2) if C2 else None
    xc = torch.cat([x1, x2 * scale2], 1) if C2 else x1
    # ---- reference
    gamma, beta = rn(C) * 0.5 + 1, rn(C) * 0.2
    ssv = rn(B, 2 * C) * 0.3 if (ss and norm in (GN_SELF, GN_SLOTS)) else None
    if norm in (GN_SELF, GN_SLOTS):
        y = F.group_norm(nhwc(xc, B, H, W), G, gamma, beta, eps=1e-5)
        if ssv is not None:
            y = y * (ssv[:, :C, None, None] + 1) + ssv[:, C:, None, None]
    elif norm == LN:
        u = F.gelu(xc) if pre_gelu else xc
        mean, var = u.mean(1, keepdim=True), u.var(1, unbiased=False, keepdim=True)
        y = nhwc((u - mean) * (var + 1e-5).rsqrt() * gamma + (beta if ln_bias else 0), B, H, W)
    else:
        y = nhwc(xc, B, H, W)
    if silu:
        y = F.silu(y)
    w = rn(Cout, C, k, k) / (C * k * k) ** 0.5
    bias = rn(Cout)
    ref = to_rows(F.conv2d(bf(y), bf(w), None, padding=k // 2))
    # ---- op
    ldc, co_off = (Cout, 0) if (pool or tw is not None) else (Cout + 32, 16)     # epilogue pooling / k_conv3s want the bare conv output
    out0 = rn(M, ldc)
    out = d(out0.clone())
    res = rn(M, ldc) if resid else None
    res_d = d(res)
    wp = d(fused.pack_conv_weights(w))
    wsl = d(torch.full((S, M, (Cout + 15) // 16 * 16), float("nan"))) if S > 1 else None
    slots_out = d(torch.full((M // 16, ldc // 16, 2), float("nan"))) if (slots and S == 1 and Cout % 16 == 0) else None
    sl1 = d(slots_of(x1, M, C1)) if norm == GN_SLOTS else None
    sl2 = d(slots_of(x2, M, C2)) if (norm == GN_SLOTS and C2) else None
    x2_d, bias_d, gamma_d, ssv_d = d(x2), d(bias), d(gamma), d(ssv)
    beta_d = d(beta) if (norm != LN or ln_bias) else None
    n_frags = (Cout + 15) // 16
    wk = rn(Cout) if (logits or pool) else None
    wk_d = d(wk)
    lpart = d(torch.full((S * n_frags, M), float("nan"))) if logits else None
    if pool:          # GlobalContext pooling in the epilogue: w_eff[tap][channel] = sum_n wk[n] * W[n][channel][tap], k-step order, bf16
        assert pipe and not logits and k == 3
        weff = torch.einsum("n,ncyx->yxc", wk, bf(w)).reshape(-1).to(torch.bfloat16).contiguous()
        wk_d = d(weff)
        lpart = d(torch.full((M // 16 * Cout + M // 16 * 2,), float("nan")))
    op = fused.mkop(OP_FCONV, (1 if silu else 0) | (2 if pre_gelu else 0) | (4 if accum else 0) | (8 if out_gelu else 0) | (32 if pipe else 0)
                    | (64 if pool else 0) | (128 if general else 0),     # 128: keep the general kernel (r05: k_conv4_gn takes the 4x4 geometry otherwise)
                    p=(s1["p"], s1["a"], s1["b"], s1["r"], sl1, x2_d, sl2, wp, bias_d, out, res_d, wsl, slots_out, gamma_d, beta_d, ssv_d, dbg, wk_d, lpart),
                    i=(B, H, W, C1, C2, Cout, ldc, co_off, k, s1["mode"], s1["groups"], s1["npad"], norm, G, TR, WM, WN, S, 2 * C)
                    + ((1,) if one_image else ((((0 if tw == W else tw) << 2) | (2 if keep_pipe else 0),) if tw is not None else ())),      # i[19] bit 0: one image per workgroup (k_conv4_gn) where k_conv4_gn_mb would take the op; bit 1 / bits 2..: k_conv3s
                    f=(1e-5, 1.0, scale2))
    ops = [op]
    if pair:          # conv1 || res_conv in one launch (k_conv_fused_pair): a 1x1 conv of the RAW concat next to the normalised 3x3 one
        w2 = rn(Cout, C, 1, 1) / C ** 0.5
        bias2 = rn(Cout)
        ref2 = to_rows(F.conv2d(bf(nhwc(xc, B, H, W)), bf(w2), None)) + bias2
        out2 = d(torch.full((M, Cout), float("nan")))
        wp2, bias2_d = d(fused.pack_conv_weights(w2)), d(bias2)
        op.flags |= 16
        ops.append(fused.mkop(OP_FCONV, 0,
                              p=(s1["p"] if not lazy else None, s1["a"], s1["b"], s1["r"], None, x2_d, None, wp2, bias2_d, out2, None, None, None,
                                 None, None, None, None, None, None),
                              i=(B, H, W, C1, C2, Cout, Cout, 0, 1, s1["mode"], s1["groups"], s1["npad"], NONE, G, TR, WM, WN, 1, 0),
                              f=(1e-5, 1.0, scale2)))
    run_ops(ops, backend)
    if pair:
        e2 = rel(out2.cpu(), ref2)
        assert e2 < tol, f"paired res_conv mismatch rel {e2}"
    if reps > 1:                                         # timing aid (tools/fconv_phases.py): the output is re-written, not checked again
        import time
        t0 = time.time()
        run_ops(ops * reps, backend)
        return (time.time() - t0) / reps
    out = out.cpu()
    if S > 1:
        got = wsl.cpu()[:, :, :Cout].sum(0)
        want = ref
    else:
        got = out[:, co_off:co_off + Cout]
        want = ref + bias
        if resid:
            want = want + res[:, co_off:co_off + Cout]
        if accum:
            want = want + out0[:, co_off:co_off + Cout]
        if out_gelu:
            want = F.gelu(want)
        # columns outside [co_off, co_off + Cout) are untouched
        assert torch.equal(out[:, :co_off], out0[:, :co_off]) and torch.equal(out[:, co_off + Cout:], out0[:, co_off + Cout:])
    assert torch.isfinite(got).all()
    e = rel(got, want)
    assert e < tol, f"conv mismatch rel {e}"
    if lazy:
        assert torch.allclose(s1["p"].cpu(), x1, atol=1e-5), "lazy source not materialised correctly"
    if logits:                                           # partial context logits: per-pixel sums equal value . wk (bias excluded when sliced)
        assert not (accum or resid)                      # the GlobalContext input is the bare conv output (bias is pixel-constant)
        val = got if S > 1 else out[:, co_off:co_off + Cout] - bias
        assert torch.allclose(lpart.cpu().sum(0), val @ wk, rtol=1e-3, atol=2e-3), "context logits wrong"
    if pool:                                             # merged over the 16-pixel chunks of an image = softmax(out . wk) pooling of out
        buf = lpart.cpu()
        part, ms = buf[:M // 16 * Cout].view(B, HW // 16, Cout), buf[M // 16 * Cout:].view(B, HW // 16, 2)
        assert torch.isfinite(buf).all()
        mx = ms[..., 0].max(1, keepdim=True).values
        wj = (ms[..., 0] - mx).exp()
        pooled = (wj[..., None] * part).sum(1) / (wj * ms[..., 1]).sum(1, keepdim=True)
        val = out[:, :Cout]
        lg = ((val - bias) @ wk).view(B, HW)
        want_pool = torch.einsum("bp,bpc->bc", torch.softmax(lg, 1), val.view(B, HW, Cout))
        ep = rel(pooled, want_pool)
        assert ep < 2e-2, f"epilogue pooling mismatch rel {ep}"
    if slots_out is not None:
        sl = slots_of(out[:, co_off:co_off + Cout].contiguous(), M, Cout)
        got_sl = slots_out.cpu()[:, co_off // 16:co_off // 16 + Cout // 16]
        if tw is not None and tw != W:                   # 2-D tiles: a fragment holds a 16-pixel patch, not 16 consecutive pixels; consumers sum per image
            sl, got_sl = sl.view(B, HW // 16, Cout // 16, 2).sum(1), got_sl.reshape(B, HW // 16, Cout // 16, 2).sum(1)
        assert torch.allclose(got_sl, sl, rtol=1e-4, atol=2e-3 * (1 if tw is None or tw == W else HW // 16) ** 0.5), "output slots wrong"
    return e


def run_slots_case(backend):
    dev = "cpu" if backend == "emu" else "cuda:0"
    g = torch.Generator().manual_seed(9)
    M, C, HW = 64, 48, 32
    h, gate, res = torch.randn(M, C, generator=g), torch.rand(2, C, generator=g), torch.randn(M, C, generator=g)
    hd, gd, rd = h.to(dev), gate.to(dev), res.to(dev)
    out, sl = torch.zeros(M, C, device=dev), torch.zeros(M // 16, C // 16, 2, device=dev)
    run_ops([fused.mkop(OP_SLOTS, p=(hd, gd, rd, out, sl), i=(M, C, HW))], backend)
    x = h * gate.repeat_interleave(HW, 0) + res
    assert torch.allclose(out.cpu(), x, atol=1e-6) and torch.allclose(sl.cpu(), slots_of(x, M, C), rtol=1e-5, atol=1e-4)
    # split-K source: x = bias + sum of slabs, materialised into `out3`
    groups, npad = 3, C + 16
    ws = torch.randn(groups, M, npad, generator=g)
    bias = torch.randn(C, generator=g)
    x3 = ws[:, :, :C].sum(0) + bias
    out3, sl3 = torch.full((M, C), float("nan"), device=dev), torch.zeros_like(sl)
    wsd, bd = ws.to(dev), bias.to(dev)
    run_ops([fused.mkop(OP_SLOTS, p=(None, None, None, out3, sl3, wsd, bd), i=(M, C, HW, groups, npad))], backend)
    assert torch.allclose(out3.cpu(), x3, atol=1e-5) and torch.allclose(sl3.cpu(), slots_of(x3, M, C), rtol=1e-5, atol=1e-4)
    sl2, xd = torch.zeros_like(sl), x.to(dev)
    run_ops([fused.mkop(OP_SLOTS, p=(xd, None, None, None, sl2), i=(M, C, HW))], backend)
    assert torch.allclose(sl2.cpu(), slots_of(x, M, C), rtol=1e-5, atol=1e-4)


def run_gca_case(backend, B, H, C, lazy=False, seed=0, epilogue_chunks=False, rc=None):
    """k_gca_pool -> k_gca_net0 -> k_gca_gate against GlobalContext + gated residual (imagen_pytorch.py:916-941, :727-729).
    rc = (C1, C2): the residual is the block's res_conv -- a 1x1 conv of the raw concat(x, skip * 2^-1/2) -- computed by an
    un-normalised fconv op that shares the pooling launch (k_gca_pool_rc, r04: flag 16 on the fconv, the pooling op right behind it)."""
    dev = "cpu" if backend == "emu" else "cuda:0"
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    HW, M, HID = H * H, B * H * H, max(3, C // 2)
    h2, res, wk = rn(M, C), rn(M, C), rn(C) * 0.3
    W0, b0, W2, b2 = rn(HID, C) / C ** 0.5, rn(HID) * 0.1, rn(C, HID) / HID ** 0.5, rn(C) * 0.1
    rc_op = None
    if rc is not None:
        assert not epilogue_chunks and H == 4
        C1, C2 = rc
        xa, xb = rn(M, C1) * 1.5 + 0.3, (rn(M, C2) if C2 else None)
        xc = torch.cat([xa, xb * 2 ** -0.5], 1) if C2 else xa
        w_rc, b_rc = rn(C, C1 + C2, 1, 1) / (C1 + C2) ** 0.5, rn(C)
        res = to_rows(F.conv2d(bf(nhwc(xc, B, H, H)), bf(w_rc), None)) + b_rc
    sm = torch.softmax((h2 @ wk).view(B, HW), 1)
    pooled = (sm[:, :, None] * h2.view(B, HW, C)).sum(1)
    hid = F.silu(pooled @ bf(W0).t() + b0)
    gate = torch.sigmoid(hid @ bf(W2).t() + b2)
    want = h2 * gate.repeat_interleave(HW, 0) + res
    nparts = C // 16
    lpart = (h2 * wk).view(M, nparts, 16).sum(-1).t().contiguous() + 3.0       # a per-pixel-constant shift must not matter
    chunks = min(8, HW // 16)
    CH = HW // chunks
    Kp, Kp2 = (C + 7) // 8 * 8, (HID + 7) // 8 * 8
    dv = lambda t: t.to(dev)
    if lazy:
        groups, npad = 3, C + 16
        ws = rn(groups, M, npad)
        bias = rn(C)
        ws[0, :, :C] = h2 - ws[1:, :, :C].sum(0) - bias
        h2_d, ws_d, bias_d = dv(torch.full((M, C), float("nan"))), dv(ws), dv(bias)
    else:
        groups = npad = 0
        h2_d, ws_d, bias_d = dv(h2), None, None
    lp_d, res_d = dv(lpart), dv(res)
    if rc is not None:
        res_d = dv(torch.full((M, C), float("nan")))          # written by the res_conv half of the shared launch
        xa_d, xb_d, wrc_d, brc_d = dv(xa), (dv(xb) if C2 else None), dv(fused.pack_conv_weights(w_rc)), dv(b_rc)      # (kept alive: ops hold raw pointers)
        rc_op = fused.mkop(OP_FCONV, 16,
                           p=(xa_d, None, None, None, None, xb_d, None, wrc_d, brc_d, res_d,
                              None, None, None, None, None, None, None, None, None),
                           i=(B, H, H, C1, C2, C, C, 0, 1, 0, 0, 0, NONE, 8, 4, 1, 1, 1, 0), f=(1e-5, 1.0, 2 ** -0.5))
    part_pool, part_ms = torch.zeros(B * chunks, C, device=dev), torch.zeros(B * chunks, 2, device=dev)
    W0p = dv(F.pad(W0, (0, Kp - C)).to(torch.bfloat16).contiguous())
    W2p = dv(F.pad(W2, (0, Kp2 - HID)).to(torch.bfloat16).contiguous())
    b0_d, b2_d = dv(b0), dv(b2)
    hid_d, out = torch.zeros(B, HID, device=dev), torch.zeros(M, C, device=dev)
    slots = torch.zeros(M // 16, C // 16, 2, device=dev)
    head = ([rc_op] if rc_op is not None else []) + [
         fused.mkop(OP_GCA, 1, p=(h2_d, ws_d, bias_d, lp_d, part_pool, part_ms), i=(M, C, HW, CH, chunks, nparts, groups, npad)),
         fused.mkop(OP_GCA, 2, p=(part_pool, part_ms, W0p, b0_d, hid_d), i=(B, C, Kp, HID, chunks))]
    if epilogue_chunks:            # the pooled 16-pixel fragments as the producing conv's epilogue leaves them (fused_pipe.h POOL):
        assert not lazy            # HW / 16 chunks per image, net0 merges up to 64 of them
        chunks = HW // 16
        lg = (h2 @ wk).view(B * chunks, 16)
        mj = lg.max(1, keepdim=True).values
        e = (lg - mj).exp()
        part_pool = dv(torch.einsum("jp,jpc->jc", e, h2.view(B * chunks, 16, C)).contiguous())
        part_ms = dv(torch.cat([mj, e.sum(1, keepdim=True)], 1).contiguous())
        head = [fused.mkop(OP_GCA, 2, p=(part_pool, part_ms, W0p, b0_d, hid_d), i=(B, C, Kp, HID, chunks))]
    ops = head + [
           fused.mkop(OP_GCA, 3, p=(h2_d, res_d, hid_d, W2p, b2_d, out, slots), i=(M, C, HW, HID, Kp2))]
    run_ops(ops, backend)
    assert torch.allclose(hid_d.cpu(), hid, rtol=2e-4, atol=2e-5), "hidden vector wrong"
    if rc is not None:
        assert rel(res_d.cpu(), res) < 4e-3, "res_conv beside the pooling launch wrong"
        want = h2 * gate.repeat_interleave(HW, 0) + res_d.cpu()
    assert torch.allclose(out.cpu(), want, rtol=2e-4, atol=2e-4), "gated residual wrong"
    assert torch.allclose(slots.cpu(), slots_of(want, M, C), rtol=1e-4, atol=2e-3)
    if lazy:
        assert torch.allclose(h2_d.cpu(), h2, atol=1e-5)


def run_attn_case(backend, B=2, cross=False, context=True, Cout=48, seed=0, tol=4e-3, dbg=None, reps=1, WN=1, general=False):
    """k_conv_fused<.., FNORM_ATTN>: the 16-token attention core (8 heads x 64, keys = [context tokens,] null k/v, the tokens' one
    shared k/v head -- or, cross-attention, null + 2 per-head time tokens; imagen_pytorch.py:480-566, :731-805) as the prologue of
    its output projection, against softmax(q k^T scale) v -> bf16 -> linear in torch."""
    dev = "cpu" if backend == "emu" else "cuda:0"
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    heads, dh, inner = 8, 64, 512
    scale = dh ** -0.5
    nq = inner if cross else inner + 2 * dh
    qkv = rn(B * 16, nq) * 1.3
    null_kv = rn(2, dh)
    q = qkv[:, :inner].view(B, 16, heads, dh)
    if cross:                                            # time block row per batch: [.. | k (2 tokens x 8 heads x 64) | v (same)]
        st = 3000
        tb = rn(B, st)
        off = 700
        kt = tb[:, off:off + 2 * inner].view(B, 2, heads, dh)
        vt = tb[:, off + 2 * inner:off + 4 * inner].view(B, 2, heads, dh)
        k = torch.cat([null_kv[0].expand(B, 1, heads, dh), kt], 1)                      # [B, 3, heads, dh]
        v = torch.cat([null_kv[1].expand(B, 1, heads, dh), vt], 1)
        tb_d, nk_d = tb.to(dev), null_kv.reshape(-1).contiguous().to(dev)
        # layout of the plan: k of token t, head h at off + t * (2 * inner) ... the plan's own strides: row = 2 * inner, head = dh, v = k + inner
        kt2 = torch.stack([kt, vt], 2).reshape(B, 2, 2 * inner)                          # [B, token, (k heads | v heads)]
        tb2 = tb.clone()
        tb2[:, off:off + 4 * inner] = kt2.reshape(B, -1)
        tb_d = tb2.to(dev)
        segs = [(nk_d.data_ptr(), nk_d.data_ptr() + dh * 4, 1, 0, 0, 0),
                (tb_d.data_ptr() + off * 4, tb_d.data_ptr() + (off + inner) * 4, 2, 2 * inner, st, dh)]
    else:
        ks = qkv[:, inner:inner + dh].view(B, 16, 1, dh).expand(B, 16, heads, dh)
        vs = qkv[:, inner + dh:].view(B, 16, 1, dh).expand(B, 16, heads, dh)
        parts_k, parts_v, segs = [], [], []
        nk_d = null_kv.reshape(-1).contiguous().to(dev)
        if context:
            st, off = 1000, 300
            tb = rn(B, st)
            ck = tb[:, off:off + 4 * dh].view(B, 2, 2, dh)                               # token t: [k | v]
            parts_k.append(ck[:, :, 0].view(B, 2, 1, dh).expand(B, 2, heads, dh))
            parts_v.append(ck[:, :, 1].view(B, 2, 1, dh).expand(B, 2, heads, dh))
            tb_d = tb.to(dev)
            segs.append((tb_d.data_ptr() + off * 4, tb_d.data_ptr() + (off + dh) * 4, 2, 2 * dh, st, 0))
        parts_k += [null_kv[0].expand(B, 1, heads, dh), ks]
        parts_v += [null_kv[1].expand(B, 1, heads, dh), vs]
        k, v = torch.cat(parts_k, 1), torch.cat(parts_v, 1)
    qkv_d = qkv.to(dev)
    if not cross:
        kp = qkv_d.data_ptr() + inner * 4
        segs += [(nk_d.data_ptr(), nk_d.data_ptr() + dh * 4, 1, 0, 0, 0), (kp, kp + dh * 4, 16, nq, 16 * nq, 0)]
    sim = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    att = torch.einsum("bhij,bjhd->bihd", torch.softmax(sim, -1), v).reshape(B * 16, inner)
    w = rn(Cout, inner, 1, 1) / inner ** 0.5
    bias, res = rn(Cout), rn(B * 16, Cout)
    want = bf(att) @ bf(w.view(Cout, inner)).t() + bias + res
    out = torch.full((B * 16, Cout), float("nan")).to(dev)
    wp, bias_d, res_d = fused.pack_conv_weights(w).to(dev), bias.to(dev), res.to(dev)
    segs = segs + [(0, 0, 0, 0, 0, 0)] * (3 - len(segs))
    op = fused.mkop(OP_FCONV, 128 if general else 0,      # 128: keep the general kernel (r06: k_lin4_attn takes the op otherwise)
                    p=(qkv_d, None, None, None, None, None, None, wp, bias_d, out, res_d, None, None, None, None, None, dbg, None, None) + tuple(sg[0] for sg in segs),
                    i=(B, 4, 4, inner, 0, Cout, Cout, 0, 1, 0, 0, 0, ATTN, 8, 4, 1, WN, 1, 0, nq) + tuple(x for sg in segs for x in sg[2:]),
                    f=(1e-5, 1.0, 1.0) + tuple((sg[1] - sg[0]) // 4 for sg in segs) + (scale,))
    run_ops([op], backend)
    if reps > 1:
        import time
        t0 = time.time()
        run_ops([op] * reps, backend)
        return (time.time() - t0) / reps
    e = rel(out.cpu(), want)
    assert torch.isfinite(out.cpu()).all() and e < tol, f"attention + projection mismatch rel {e}"
    return e


ATTN_CASES = {"self_context": dict(B=2, cross=False, context=True, seed=61), "self_plain": dict(B=1, cross=False, context=False, seed=62),
              "cross_time_tokens": dict(B=2, cross=True, seed=63, Cout=64),
              "self_context_wn2": dict(B=2, cross=False, context=True, seed=68, Cout=64, WN=2),      # the B >= 8 tile of the output projection
              # r06: the cases above run on k_lin4_attn (csrc/fused_conv4.h); op flag 128 keeps k_conv_fused<.., FNORM_ATTN>
              "self_context_on_the_general_kernel": dict(B=2, cross=False, context=True, seed=61, general=True),
              "cross_time_tokens_on_the_general_kernel": dict(B=2, cross=True, seed=63, Cout=64, general=True),
              "self_plain_ragged_cout": dict(B=3, cross=False, context=False, seed=69, Cout=40)}

GCA_CASES = {"4x4_lazy": dict(B=2, H=4, C=128, lazy=True),
             "4x4_lazy_res_conv_beside_pool": dict(B=2, H=4, C=128, lazy=True, seed=21, rc=(96, 64)),
             "4x4_res_conv_beside_pool_no_skip": dict(B=1, H=4, C=64, seed=22, rc=(128, 0)), "8x8": dict(B=1, H=8, C=64, seed=1), "16x16": dict(B=1, H=16, C=64, seed=2),
             "8x8_hid128_gate_t": dict(B=2, H=8, C=256, seed=31),
             "32x32_c256_64_chunks_net0_t": dict(B=1, H=32, C=256, seed=32, epilogue_chunks=True),      # k_gca_net0_t<256, 64>
             "16x16_c512_16_chunks_net0_t": dict(B=1, H=16, C=512, seed=33, epilogue_chunks=True),      # k_gca_net0_t<512, 16>, k_gca_gate_t<256>       # r05: HID = 128 | 256 | 512 run k_gca_gate_t (compile-time hidden width)
             "8x8_c192_b2": dict(B=2, H=8, C=192, seed=8),
             "32x32_epilogue_chunks": dict(B=2, H=32, C=64, seed=11, epilogue_chunks=True),
             "16x16_epilogue_chunks": dict(B=1, H=16, C=320, seed=12, epilogue_chunks=True),
             "8x8_epilogue_chunks": dict(B=2, H=8, C=128, seed=13, epilogue_chunks=True)}
GCA_CASES_FULL = {"unet_32x32_epilogue_chunks": dict(B=1, H=32, C=256, seed=14, epilogue_chunks=True),
                  "unet_b4_16x16_epilogue_chunks": dict(B=4, H=16, C=512, seed=15, epilogue_chunks=True),
                  "unet_4x4": dict(B=1, H=4, C=1024, lazy=True, seed=3),
                  "unet_4x4_res_conv_beside_pool": dict(B=1, H=4, C=1024, lazy=True, seed=23, rc=(1024, 1024)),
                  "unet_b4_4x4_res_conv_beside_pool": dict(B=4, H=4, C=1024, lazy=True, seed=24, rc=(1024, 1024)), "unet_8x8": dict(B=1, H=8, C=1024, seed=4),
                  "unet_32x32": dict(B=1, H=32, C=256, seed=5), "unet_b4_16x16": dict(B=4, H=16, C=512, seed=6)}


# name -> kwargs of run_conv_case; small enough for the CPU-thread emulator, and every kernel path is in here
CONV_CASES = {
    # the 4x4 level: whole image per tile, input-channel slices, lazy split-K source, partial slabs out
    "gn_self_sliced_lazy_splitk_4x4": dict(B=2, H=4, W=4, C1=64, C2=0, Cout=48, k=3, norm=GN_SELF, WM=1, WN=1, S=2, lazy=1, logits=True),
    # 12 channels per group (not a power of two), 24 channel chunks per pixel (512 threads % 24 != 0): the fixed-order fallback
    # of the GN_SELF statistics (r02: LDS float atomics, order dependent)
    "gn_self_odd_groups_fallback_4x4": dict(B=2, H=4, W=4, C1=96, C2=0, Cout=32, k=3, norm=GN_SELF, WM=1, WN=1, seed=41),
    "gn_self_concat_gate_lazy_4x4": dict(B=1, H=4, W=4, C1=64, C2=64, Cout=32, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=2, seed=1),
    # r05: the geometry k_conv4_gn is written for (csrc/fused_conv4.h: 4 slices of two whole groups each, Cs = 256 | 512), every lazy
    # mode, with and without scale / shift and SiLU, partial context logits, two images; `general=True` = the same op on k_conv_fused
    "conv4_gn_lazy_splitk_1024": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=32, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=71, logits=True),
    "conv4_gn_concat_gate_2048": dict(B=1, H=4, W=4, C1=1024, C2=1024, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=2, seed=72),
    "conv4_gn_plain_1024_b2": dict(B=2, H=4, W=4, C1=1024, C2=0, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, ss=False, silu=False, seed=73),
    "conv4_gn_concat_lazy_2048": dict(B=1, H=4, W=4, C1=1024, C2=1024, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=74, logits=True),
    # r05: B >= 2 in that geometry runs k_conv4_gn_mb -- NB = 2 | 4 images per workgroup share its weight slice (B % 4 == 0 at Cs = 256: 4, other
    # even B: 2; Cs = 512: 2); every lazy mode (a split-K element's six loads bound how many images' gathers fly together: 2 at Cs = 256, 1 at
    # Cs = 512), several image groups per (slice, n-tile), logits; `one_image=True` = the same op on k_conv4_gn
    "conv4_mb_lazy_splitk_1024_b4": dict(B=4, H=4, W=4, C1=1024, C2=0, Cout=32, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=75, logits=True),
    "conv4_mb_gate_1024_b8": dict(B=8, H=4, W=4, C1=1024, C2=0, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=2, seed=76),
    "conv4_mb_lazy_splitk_1024_b6": dict(B=6, H=4, W=4, C1=1024, C2=0, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=77, logits=True),
    "conv4_mb_concat_lazy_2048_b2": dict(B=2, H=4, W=4, C1=1024, C2=1024, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=78, logits=True),
    "conv4_mb_concat_gate_2048_b4": dict(B=4, H=4, W=4, C1=1024, C2=1024, Cout=32, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=2, seed=79),
    "conv4_mb_concat_plain_2048_b2": dict(B=2, H=4, W=4, C1=1024, C2=1024, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, seed=80, ss=False),
    "conv4_mb_concat_lazy_512_512_b4": dict(B=4, H=4, W=4, C1=512, C2=512, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=81),
    "conv4_one_image_per_workgroup_b4": dict(B=4, H=4, W=4, C1=1024, C2=0, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=75, one_image=True),
    # r06: the same geometry without a norm (k_conv4_gn<64, 0, false>): the merged 3x3 + 1x1 conv of the last Downsample at B = 1
    "conv4_no_norm_plain_1024": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=32, k=3, norm=NONE, WM=1, WN=1, S=4, silu=False, ss=False, slots=False, seed=145),
    "conv4_geometry_on_the_general_kernel": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=16, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=71, general=True),
    # 8x8 level: 2-row tiles with halo rows from neighbouring tiles, statistics from producer slots, final epilogue + slots
    "gn_slots_concat_8x8": dict(B=1, H=8, W=8, C1=128, C2=128, Cout=32, k=3, norm=GN_SLOTS, WM=1, WN=1, resid=True, seed=2),
    # 32-pixel rows (WM = 2), 2 n-fragments per tile, accumulate mode
    "gn_slots_wide_rows_wm2_wn2": dict(B=2, H=4, W=32, C1=128, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=2, WN=2, accum=True, seed=3),
    # XCD-aware tile order (8 n-tiles)
    "gn_slots_xcd_map_16x16": dict(B=1, H=16, W=16, C1=128, C2=0, Cout=128, k=3, norm=GN_SLOTS, WM=1, WN=1, ss=False, seed=4, logits=True),
    "raw_1x1_concat_res_conv": dict(B=1, H=8, W=8, C1=64, C2=32, Cout=40, k=1, norm=NONE, WM=1, WN=1, silu=False, slots=False, seed=5),
    # conv1 || res_conv pairs (one launch): slot statistics at 8x8, and the 4x4 level with a lazy split-K source read by both halves
    "pair_gn_slots_concat_8x8": dict(B=1, H=8, W=8, C1=64, C2=64, Cout=32, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=21, pair=True),
    "pair_gn_self_lazy_splitk_4x4": dict(B=2, H=4, W=4, C1=64, C2=64, Cout=48, k=3, norm=GN_SELF, WM=1, WN=1, S=2, lazy=1, seed=22, pair=True),
    "pair_gn_slots_wm2_wn2": dict(B=1, H=2, W=32, C1=128, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=23, pair=True),
    # pipelined kernel (k_conv_fused_pipe): 8x8 with halo rows and a concat that changes source inside a chunk's thread range,
    # 16-pixel rows with 2 n-fragments + logits, 32-pixel rows, accumulate + residual
    "pipe_gn_slots_concat_8x8": dict(B=1, H=8, W=8, C1=128, C2=128, Cout=32, k=3, norm=GN_SLOTS, WM=1, WN=1, resid=True, seed=31, pipe=True),
    "pipe_gn_slots_16x16_wn2": dict(B=1, H=16, W=16, C1=256, C2=128, Cout=64, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=32, pipe=True, logits=True),
    "pipe_gn_slots_wm2_wn2_accum": dict(B=2, H=4, W=32, C1=128, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=2, WN=2, accum=True, seed=33, pipe=True),
    "pipe_pair_gn_slots_concat_8x8": dict(B=1, H=8, W=8, C1=128, C2=128, Cout=32, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=38, pipe=True, pair=True),
    # GlobalContext pooling in the conv's epilogue (context logits from the conv's own input through w_eff)
    "pipe_pool_8x8": dict(B=2, H=8, W=8, C1=128, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=54, pipe=True, pool=True),
    "pipe_pool_16x16_wn2": dict(B=1, H=16, W=16, C1=256, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=55, pipe=True, pool=True),
    "pipe_pool_16x16_tr2": dict(B=2, H=16, W=16, C1=128, C2=0, Cout=96, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=56, pipe=True, pool=True),
    "pipe_pool_wide_rows_wm4": dict(B=2, H=4, W=32, C1=128, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=57, pipe=True, pool=True),
    # 32-pixel tiles of the 16x16 / 8x8 maps (the B >= 4 geometry): two row pairs / four rows per workgroup
    "pipe_gn_slots_16x16_tr2_wn2": dict(B=2, H=16, W=16, C1=256, C2=128, Cout=64, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=42, pipe=True, logits=True),
    "pipe_gn_slots_wide_rows_wm4_wn2": dict(B=2, H=4, W=32, C1=128, C2=128, Cout=64, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=47, pipe=True, logits=True),
    "pipe_pair_gn_slots_wide_rows_wm4": dict(B=1, H=4, W=32, C1=128, C2=0, Cout=64, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=48, pipe=True, pair=True),
    "gn_slots_wide_rows_wm4_wn1": dict(B=1, H=4, W=32, C1=128, C2=0, Cout=48, k=3, norm=GN_SLOTS, WM=4, WN=1, seed=49),
    "pipe_gn_slots_8x8_wn2": dict(B=2, H=8, W=8, C1=128, C2=128, Cout=64, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=46, pipe=True),
    "pipe_gn_slots_8x8_tr4_resid": dict(B=2, H=8, W=8, C1=128, C2=128, Cout=48, k=3, norm=GN_SLOTS, WM=2, WN=1, resid=True, seed=43, pipe=True),
    "pipe_pair_gn_slots_16x16_tr2": dict(B=2, H=16, W=16, C1=128, C2=128, Cout=64, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=44, pipe=True, pair=True),
    "raw_1x1_res_conv_8x8_tr4": dict(B=2, H=8, W=8, C1=64, C2=64, Cout=32, k=1, norm=NONE, WM=2, WN=2, silu=False, slots=False, seed=45),
    # r05: the shapes k_lin4_ln takes (csrc/fused_conv4.h: plain source of 1024 | 2048 channels on the 16-token map): ff1 with its GELU
    # epilogue, ff2 with residual and two n-fragments per tile, a bias-carrying LayerNorm, two images
    "lin4_ln_ff1_1024_gelu": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=48, k=1, norm=LN, WM=1, WN=1, silu=False, out_gelu=True, seed=81),
    "lin4_ln_ff2_2048_resid_wn2": dict(B=2, H=4, W=4, C1=2048, C2=0, Cout=64, k=1, norm=LN, WM=1, WN=2, silu=False, resid=True, seed=82),
    "lin4_ln_bias_1024_b2": dict(B=2, H=4, W=4, C1=1024, C2=0, Cout=32, k=1, norm=LN, WM=1, WN=1, silu=False, ln_bias=True, accum=True, seed=83),
    "lin4_shape_on_the_general_kernel": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=48, k=1, norm=LN, WM=1, WN=1, silu=False, out_gelu=True, seed=81, general=True),
    "layernorm_linear": dict(B=2, H=4, W=4, C1=128, C2=0, Cout=64, k=1, norm=LN, WM=1, WN=1, silu=False, seed=6, out_gelu=True),
    "layernorm_lazy_splitk_linear": dict(B=2, H=4, W=4, C1=128, C2=0, Cout=64, k=1, norm=LN, WM=1, WN=1, silu=False, lazy=1, seed=7),
    "gelu_layernorm_bias_linear": dict(B=2, H=4, W=4, C1=128, C2=0, Cout=64, k=1, norm=LN, WM=1, WN=2, silu=False, pre_gelu=True,
                                       ln_bias=True, seed=6),
}
# the UNet's own layer shapes (B = 1): GPU only
CONV_CASES_FULL = {
    "unet_4x4_1024_s4": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=10),
    "unet_4x4_2048_s4_gate": dict(B=1, H=4, W=4, C1=1024, C2=1024, Cout=1024, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=2, seed=11),
    "unet_8x8_1536": dict(B=1, H=8, W=8, C1=1024, C2=512, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=12),
    "unet_16x16_768": dict(B=1, H=16, W=16, C1=512, C2=256, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, resid=True, seed=13),
    "unet_32x32_512": dict(B=1, H=32, W=32, C1=256, C2=256, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=14),
    "unet_32x32_res_conv": dict(B=1, H=32, W=32, C1=256, C2=256, Cout=256, k=1, norm=NONE, WM=2, WN=2, silu=False, seed=15),
    "unet_ln_ff2_2048": dict(B=1, H=4, W=4, C1=2048, C2=0, Cout=1024, k=1, norm=LN, WM=1, WN=1, silu=False, pre_gelu=True, resid=True, seed=17),
    # the LayerNorm linears as the r04 plan runs them (plain sources): ff1 (+ GELU epilogue), ff2, the merged q | k | v projection
    "unet_ln_ff1_1024": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=2048, k=1, norm=LN, WM=1, WN=1, silu=False, out_gelu=True, seed=65),
    "unet_ln_ff2_2048_plain": dict(B=1, H=4, W=4, C1=2048, C2=0, Cout=1024, k=1, norm=LN, WM=1, WN=1, silu=False, resid=True, seed=66),
    "unet_ln_qkv_1024": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=640, k=1, norm=LN, WM=1, WN=1, silu=False, seed=67),
    "unet_ln_qkv_lazy": dict(B=1, H=4, W=4, C1=1024, C2=0, Cout=640, k=1, norm=LN, WM=1, WN=1, silu=False, lazy=1, seed=18),
    "unet_pair_8x8_1536": dict(B=1, H=8, W=8, C1=1024, C2=512, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=24, pair=True),
    "unet_pair_4x4_2048_lazy": dict(B=1, H=4, W=4, C1=1024, C2=1024, Cout=1024, k=3, norm=GN_SELF, WM=1, WN=1, S=4, lazy=1, seed=25, pair=True),
    "unet_pair_32x32_512": dict(B=1, H=32, W=32, C1=256, C2=256, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=26, pair=True),
    "unet_pipe_8x8_1536": dict(B=1, H=8, W=8, C1=1024, C2=512, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=34, pipe=True),
    "unet_pipe_16x16_768": dict(B=1, H=16, W=16, C1=512, C2=256, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, resid=True, seed=35, pipe=True),
    "unet_pipe_32x32_512": dict(B=1, H=32, W=32, C1=256, C2=256, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=36, pipe=True),
    "unet_pipe_32x32_256": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=37, pipe=True, logits=True),
    "unet_pipe_pair_16x16_768": dict(B=1, H=16, W=16, C1=512, C2=256, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=39, pipe=True, pair=True),
    # B = 4 geometry (r03): 32-pixel tiles at 8x8 / 16x16, 64-pixel tiles at 32x32; the 1536-channel frame only fits the chunked
    # (pipelined) kernel
    "unet_b4_pipe_pair_8x8_1536": dict(B=4, H=8, W=8, C1=1024, C2=512, Cout=1024, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=50, pipe=True, pair=True),
    "unet_b4_pipe_16x16_768": dict(B=4, H=16, W=16, C1=512, C2=256, Cout=512, k=3, norm=GN_SLOTS, WM=2, WN=2, resid=True, seed=51, pipe=True),
    "unet_b4_pipe_32x32_512": dict(B=4, H=32, W=32, C1=256, C2=256, Cout=256, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=52, pipe=True, logits=True),
    "unet_b4_pipe_pair_32x32_512": dict(B=4, H=32, W=32, C1=256, C2=256, Cout=256, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=53, pipe=True, pair=True),
    "unet_pipe_pool_32x32_256": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=58, pipe=True, pool=True),
    "unet_pipe_pool_8x8_1024": dict(B=1, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=59, pipe=True, pool=True),
    "unet_b4_pipe_pool_16x16_512": dict(B=4, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=60, pipe=True, pool=True),
    "unet_b4_4x4": dict(B=4, H=4, W=4, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SELF, WM=1, WN=1, S=1, seed=16),
    # r06: k_conv3s (csrc/fused_conv3s.h), the recurring single-source geometries with compile-time shapes: full-width strips (tw = W) and
    # 2-D tiles (tw < W), with / without residual, scale-shift, epilogue pooling; `keep_pipe` = the same op on the general pipelined kernel
    "conv3s_32x32_256_strip": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=90, pipe=True, tw=32),
    "conv3s_32x32_256_tile4x8_resid": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=91, pipe=True, tw=8, resid=True),
    "conv3s_32x32_256_tile4x8_pool": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=92, pipe=True, tw=8, pool=True),
    "conv3s_32x32_256_strip_pool_no_ss": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=93, pipe=True, tw=32, pool=True, ss=False),
    "conv3s_16x16_256_strip": dict(B=2, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=94, pipe=True, tw=16, resid=True),
    "conv3s_16x16_256_tile4x4_pool": dict(B=1, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=95, pipe=True, tw=4, pool=True),
    "conv3s_16x16_512_tile4x4": dict(B=1, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=96, pipe=True, tw=4, resid=True),
    "conv3s_16x16_512_strip_pool": dict(B=1, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=97, pipe=True, tw=16, pool=True),
    "conv3s_8x8_1024_resid": dict(B=1, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=104, pipe=True, tw=8, resid=True),
    "conv3s_8x8_512_pool": dict(B=1, H=8, W=8, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=105, pipe=True, tw=8, pool=True),
    "conv3s_16x16_512_tile4x4_pool": dict(B=1, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=103, pipe=True, tw=4, pool=True),
    "conv3s_8x8_512": dict(B=2, H=8, W=8, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=98, pipe=True, tw=8, resid=True),
    "conv3s_8x8_1024_pool": dict(B=1, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=99, pipe=True, tw=8, pool=True),
    "conv3s_b2_32x32_256_tile8x8_wn2": dict(B=2, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=100, pipe=True, tw=8, pool=True),
    "conv3s_16x16_256_strip_pool": dict(B=1, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=106, pipe=True, tw=16, pool=True),
    "conv3s_16x16_256_tile4x4": dict(B=2, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=107, pipe=True, tw=4, resid=True, ss=False),
    "conv3s_16x16_512_strip": dict(B=1, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=108, pipe=True, tw=16),
    "conv3s_b2_32x32_256_strip2_wn2": dict(B=2, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=109, pipe=True, tw=32, resid=True),
    "conv3s_b2_32x32_256_strip2_wn2_pool": dict(B=2, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=110, pipe=True, tw=32, pool=True),
    "conv3s_b3_32x32_256_tile8x8_wn2": dict(B=3, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=4, WN=2, seed=111, pipe=True, tw=8),
    # B >= 2 tiles at 16x16 (8 rows x 4 columns) and 8x8 (4 rows)
    "conv3s_b2_16x16_256_tile8x4": dict(B=2, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=1, seed=130, pipe=True, tw=4, resid=True),
    "conv3s_b2_16x16_256_tile8x4_pool": dict(B=2, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=1, seed=131, pipe=True, tw=4, pool=True),
    "conv3s_b4_16x16_256_tile8x4_wn2": dict(B=4, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=132, pipe=True, tw=4),
    "conv3s_b4_16x16_256_tile8x4_wn2_pool": dict(B=4, H=16, W=16, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=133, pipe=True, tw=4, pool=True, ss=False),
    "conv3s_b2_16x16_512_tile8x4": dict(B=2, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=134, pipe=True, tw=4, resid=True),
    "conv3s_b2_16x16_512_tile8x4_pool": dict(B=2, H=16, W=16, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=135, pipe=True, tw=4, pool=True),
    "conv3s_b2_8x8_512_rows4": dict(B=2, H=8, W=8, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=2, WN=1, seed=136, pipe=True, tw=8),
    "conv3s_b2_8x8_512_rows4_pool": dict(B=2, H=8, W=8, C1=512, C2=0, Cout=512, k=3, norm=GN_SLOTS, WM=2, WN=1, seed=137, pipe=True, tw=8, pool=True),
    "conv3s_b2_8x8_1024_rows4": dict(B=2, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=2, WN=1, seed=138, pipe=True, tw=8, resid=True),
    "conv3s_b2_8x8_1024_rows4_pool": dict(B=2, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=2, WN=1, seed=139, pipe=True, tw=8, pool=True),
    "conv3s_b4_8x8_1024_rows4_wn2": dict(B=4, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=140, pipe=True, tw=8),
    "conv3s_b4_8x8_1024_rows4_wn2_pool": dict(B=4, H=8, W=8, C1=1024, C2=0, Cout=1024, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=141, pipe=True, tw=8, pool=True),
    # r06: k_conv3s_rc -- conv1 on the concat of two sources + the block's res_conv in the same workgroups (the pipelined pairs of the B = 1 plan)
    "conv3s_rc_32x32_512_tile4x8": dict(B=2, H=32, W=32, C1=256, C2=256, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=121, pipe=True, pair=True, tw=8, ss=False),
    "conv3s_rc_16x16_768_tile4x4": dict(B=1, H=16, W=16, C1=512, C2=256, Cout=512, k=3, norm=GN_SLOTS, WM=1, WN=2, seed=123, pipe=True, pair=True, tw=4),
    "conv3s_rc_8x8_1536": dict(B=1, H=8, W=8, C1=1024, C2=512, Cout=1024, k=3, norm=GN_SLOTS, WM=1, WN=1, seed=124, pipe=True, pair=True, tw=8, ss=False),
    "conv3s_geometry_on_the_general_kernel": dict(B=1, H=32, W=32, C1=256, C2=0, Cout=256, k=3, norm=GN_SLOTS, WM=2, WN=2, seed=90, pipe=True, tw=32, keep_pipe=True),
}
