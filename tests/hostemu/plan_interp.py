"""CPU interpreter of a UNet launch plan (sparsefusion_amd/unet.py::_Plan built on host memory): checks the PLANNER --
operand wiring, lazy tensors, workspaces, slot tables -- without a GPU.  The fused ops (FCONV / SLOTS / GCA) run the
product's own kernel source on CPU threads (hostemu/fused.py); the first-round ops, whose kernels are validated on the
GPU by tests/test_gpu_unet_ops.py, are restated here in torch from their operand contracts (csrc/unet_ops.hip).
Test infrastructure only."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import fused

(OP_CONV, OP_GN_ACT, OP_LN, OP_GEMV, OP_ATTN, OP_GCA_POOL, OP_ELTWISE, OP_MEMSET, OP_TIME_EMB, OP_SPLITK_REDUCE) = range(1, 11)
OP_FCONV, OP_SLOTS, OP_GCA = 14, 15, 16


def f32(ptr, *shape):
    n = 1
    for s in shape:
        n *= s
    return torch.frombuffer((C.c_float * n).from_address(ptr), dtype=torch.float32).view(*shape)


def bf16(ptr, *shape):
    n = 1
    for s in shape:
        n *= s
    return torch.frombuffer((C.c_int16 * n).from_address(ptr), dtype=torch.bfloat16).view(*shape)


def unpack_weights(ptr, Cout, Cin, taps):
    """inverse of sf_conv_pack_weights: -> fp32 [Cout, taps, Cin]"""
    nfr, cch = (Cout + 15) // 16, Cin // 32
    x = bf16(ptr, nfr, taps, cch, 4, 16, 8).float()                 # [nf, tap, cc, kb, n16, j]
    w = x.permute(0, 4, 1, 2, 3, 5).reshape(nfr * 16, taps, Cin)
    return w[:Cout]


def lazy_value(o, ibase, M, C, B, dst):
    """LazySrc of an op (csrc/unet_ops.hip): evaluate and materialise into dst [M, C]."""
    mode, groups, npad = o.i[ibase], o.i[ibase + 1], o.i[ibase + 2]
    if mode == 0:
        return dst
    if mode == 1:
        v = f32(o.p[8], groups, M, npad)[:, :, :C].sum(0)
        if o.p[9]:
            v = v + f32(o.p[9], C)
        if o.p[10]:
            v = v + f32(o.p[10], M, C)
    else:
        h, gate = f32(o.p[8], M, C), f32(o.p[9], B, C)
        r = f32(o.p[10], M, C) if o.p[10] else dst
        v = h * gate.repeat_interleave(M // B, 0) + r
    dst.copy_(v)
    return dst


def act(v, code):
    return F.silu(v) if code == 1 else (torch.sigmoid(v) if code == 2 else v)


def run_op(o):
    t, fl, i, f, p = o.type, o.flags, o.i, o.f, o.p
    if t in (OP_FCONV, OP_SLOTS, OP_GCA):
        fused.run([o])
    elif t == OP_MEMSET:
        C.memset(p[0], 0, (i[0] & 0xffffffff) * 4)
    elif t == OP_TIME_EMB:
        B, half = i[0], i[1]
        tt, w, out = f32(p[0], B), f32(p[1], half), f32(p[3], B, 1 + 2 * half)
        fr = tt[:, None] * w[None] * 2.0 * 3.14159265358979323846
        out.copy_(torch.cat([tt[:, None], fr.sin(), fr.cos()], 1))
    elif t == OP_GEMV:
        M, N, K, Kp, ldx, ldy = i[0], i[1], i[2], i[3], i[4], i[5]
        x = f32(p[0], M, ldx)[:, :K] if M == 1 else torch.stack([f32(p[0] + 4 * m * ldx, K) for m in range(M)])
        if fl & 1:
            x = F.silu(x)
        W = bf16(p[1], N, Kp).float()[:, :K]
        y = x @ W.t()
        if p[2]:
            y = y + f32(p[2], N)
        y = act(y, (fl >> 1) & 3)
        for m in range(M):
            f32(p[3] + 4 * m * ldy, N).copy_(y[m])
    elif t == OP_LN:
        R, Cc = i[0], i[1]
        x = f32(p[0], R, Cc)
        if fl & 1:
            x = F.gelu(x)
        mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
        y = (x - mean) * (var + f[0]).rsqrt() * f32(p[1], Cc)
        if p[2]:
            y = y + f32(p[2], Cc)
        if fl & 2:
            if p[4]:
                y = y + f32(p[4], R, Cc)
            f32(p[3], R, Cc).copy_(y)
        else:
            bf16(p[3], R, Cc).copy_(y.to(torch.bfloat16))
    elif t == OP_SPLITK_REDUCE:
        M, Cout, npad, groups = i[0], i[1], i[2], i[3]
        v = f32(p[0], groups, M, npad)[:, :, :Cout].sum(0)
        if p[1]:
            v = v + f32(p[1], Cout)
        if p[2]:
            v = v + f32(p[2], M, Cout)
        f32(p[3], M, Cout).copy_(v)
    elif t == OP_ELTWISE:
        if fl == 1:
            B, HW, Cc = i[0], i[1], i[2]
            h, gate, out = f32(p[0], B * HW, Cc), f32(p[1], B, Cc), f32(p[3], B * HW, Cc)
            r = f32(p[2], B * HW, Cc) if p[2] else out
            out.copy_(h * gate.repeat_interleave(HW, 0) + r)
        elif fl == 2:
            B, HW, Cc_, Cx, Cp = i[0], i[1], i[2], i[3], i[4]
            cond, x, out = f32(p[0], B, Cc_, HW), f32(p[1], B, Cx, HW), f32(p[3], B, HW, Cp)
            out.zero_()
            out[:, :, :Cc_] = cond.permute(0, 2, 1)
            out[:, :, Cc_:Cc_ + Cx] = x.permute(0, 2, 1)
        elif fl == 3:
            B, HW, Cc, ldi = i[0], i[1], i[2], i[3]
            f32(p[3], B, Cc, HW).copy_(f32(p[0], B, HW, ldi)[:, :, :Cc].permute(0, 2, 1))
        else:
            raise NotImplementedError(f"eltwise mode {fl}")
    elif t == OP_GN_ACT:
        B, HW, C1, C2, ss_stride = i[0], i[1], i[2], i[3], i[4]
        G = i[8] or 8
        Cc, M = C1 + C2, B * HW
        s1 = lazy_value(o, 5, M, C1, B, f32(p[0], M, C1))
        x = torch.cat([s1, f32(p[1], M, C2) * f[1]], 1) if C2 else s1
        y = F.group_norm(x.view(B, HW, Cc).permute(0, 2, 1), G, f32(p[2], Cc), f32(p[3], Cc), eps=f[0])
        if p[4]:
            ss = torch.stack([f32(p[4] + 4 * b * ss_stride, 2 * Cc) for b in range(B)])
            y = y * (ss[:, :Cc, None] + 1) + ss[:, Cc:, None]
        if not fl & 1:
            y = F.silu(y)
        bf16(p[5], B, HW, Cc).copy_(y.permute(0, 2, 1).to(torch.bfloat16))
        if p[6]:
            bf16(p[6], M, Cc).copy_(x.to(torch.bfloat16))
    elif t == OP_GCA_POOL:
        B, HW, Cc = i[0], i[1], i[2]
        h = lazy_value(o, 3, B * HW, Cc, B, f32(p[0], B * HW, Cc)).view(B, HW, Cc)
        logit = h @ f32(p[1], Cc) + f32(p[2], 1)
        f32(p[4], B, HW).copy_(logit)
        sm = torch.softmax(logit, 1)
        f32(p[3], B, Cc).add_((sm[:, :, None] * h).sum(1))
    elif t == OP_ATTN:
        B, heads, ldq = i[0], i[1], i[2]
        q = f32(p[0], B * 16, ldq)
        out = f32(p[1], B * 16, heads * 64) if fl & 1 else bf16(p[1], B * 16, heads * 64)
        for b in range(B):
            for hd in range(heads):
                qq = q[b * 16:(b + 1) * 16, hd * 64:(hd + 1) * 64] * f[0]
                ks, vs = [], []
                for sgi in range(3):
                    rows, rs, bs, hs = i[4 + 4 * sgi:8 + 4 * sgi]
                    for r in range(rows):
                        off = 4 * (b * bs + r * rs + hd * hs)
                        ks.append(f32(p[2 + 2 * sgi] + off, 64))
                        vs.append(f32(p[3 + 2 * sgi] + off, 64))
                K, V = torch.stack(ks), torch.stack(vs)
                a = torch.softmax(qq @ K.t(), 1) @ V
                out[b * 16:(b + 1) * 16, hd * 64:(hd + 1) * 64] = a if fl & 1 else a.to(torch.bfloat16)
    elif t == OP_CONV:
        B, H, W, Cin, Ho, Wo, Cout, ldc, co_off, kh, kw, stride, pad, groups = [i[k] for k in range(14)]
        M = B * Ho * Wo
        ups = 1 if fl & 16 else 0
        Hs, Ws = H >> ups, W >> ups
        x = (f32(p[0], B, Hs, Ws, Cin) if fl & 1 else bf16(p[0], B, Hs, Ws, Cin).float()).permute(0, 3, 1, 2)
        if ups:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = x.to(torch.bfloat16).float()
        w = unpack_weights(p[1], Cout, Cin, kh * kw).view(Cout, kh, kw, Cin).permute(0, 3, 1, 2)
        # explicit output size: pad right / bottom as needed (the kernel's out-of-image rule)
        need_h, need_w = (Ho - 1) * stride + kh, (Wo - 1) * stride + kw
        xp = F.pad(x, (pad, max(0, need_w - W - pad), pad, max(0, need_h - H - pad)))
        y = F.conv2d(xp, w, None, stride=stride)[:, :, :Ho, :Wo].permute(0, 2, 3, 1).reshape(M, Cout)
        if groups > 1 and fl & 8:                                # deferred split-K: the sum lives in the slabs
            npad = (Cout + 15) // 16 * 16
            ws = f32(p[5], groups, M, npad)
            ws.zero_()
            ws[0, :, :Cout] = y
            return
        if p[2]:
            y = y + f32(p[2], Cout)
        if fl & 2:                                               # SiLU + PixelShuffle(2)
            y = F.silu(y).view(B, Ho, Wo, Cout // 4, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B * 4 * Ho * Wo, Cout // 4)
            f32(p[3], B * 4 * Ho * Wo, ldc)[:, co_off:co_off + Cout // 4] = y
            if p[7]:                                             # output slots: any fragment partition with the right (image, column) sums serves
                Mo, Co = B * 4 * Ho * Wo, Cout // 4
                t = y.reshape(Mo // 16, 16, Co // 16, 16).permute(0, 2, 1, 3).reshape(Mo // 16, Co // 16, 256)
                f32(p[7], Mo // 16, ldc // 16, 2)[:, co_off // 16:co_off // 16 + Co // 16] = torch.stack([t.sum(-1), (t * t).sum(-1)], -1)
            return
        if fl & 256:                                             # split-K reduction straight into the plan's NCHW output
            f32(p[3], B, Cout, Ho * Wo)[:] = y.view(B, Ho * Wo, Cout).permute(0, 2, 1)
            return
        out = f32(p[3], M, ldc)
        if p[4]:
            y = y + f32(p[4], M, ldc)[:, co_off:co_off + Cout]
        if fl & 4:
            y = y + out[:, co_off:co_off + Cout]
        if fl & 32:
            y = F.relu(y)
        if fl & 64:
            y = F.gelu(y)
        out[:, co_off:co_off + Cout] = y
    else:
        raise NotImplementedError(f"op type {t}")


def run_plan(ops, progress=False):
    k = 0
    while k < len(ops):
        o = ops[k]
        if progress and k % 20 == 0:
            print(f"  op {k}/{len(ops)}", flush=True)
        if o.type == OP_FCONV and (o.flags & 16):          # conv1 || res_conv: one launch for this op and the next
            fused.run([o, ops[k + 1]])
            k += 2
            continue
        run_op(o)
        k += 1
