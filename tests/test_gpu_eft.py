"""EFT pre-pass (row E1) on the HIP plan: kernel-level checks against plain torch, then the whole module against the
goldens of the REAL reference module (tests/golden/make_golden_eft.py) and the oracle.  bf16 MFMA operands through
the resnet trunk and 12 transformer layers: f3 (unit-variance LayerNorm output space) within 3e-2 relative L2 and
cosine > 0.999, rgb (sigmoid output) within 1e-2 absolute; measured values are printed."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import eft_ref
from eft_common import GOLD, rel_err, scene, spec, state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_KEEP = []


def _op(type_, flags, p=(), i=(), f=()):
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


def _run(*ops):
    from sparsefusion_amd import _lib
    arr = (_lib.SfOp * len(ops))(*ops)
    _lib.check(_lib.lib().sf_plan_run(arr, len(ops), _lib.stream_ptr()), "plan")
    torch.cuda.synchronize()


def _i64(v):
    return ((v & 0xffffffff) - (1 << 32) if (v & 0xffffffff) >= (1 << 31) else (v & 0xffffffff), v >> 32)


def test_pool3_resize_gather_harmonic_kernels():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 17, 17, generator=g)                       # odd size: border windows
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.empty(2, 9, 9, 64, device=DEV)
    _run(_op(11, 2, p=(xd, None, None, out), i=(2, 17, 17, 64)))
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1))
    # align-corners bilinear resize into a channel slice
    wide = torch.zeros(2, 32, 32, 96, device=DEV)
    _run(_op(13, 0, p=(xd, None, None, wide), i=(2, 17, 17, 64, 32, 32, 96, 32)))
    ref = F.interpolate(x, (32, 32), mode='bilinear', align_corners=True)
    got = wide.cpu()[..., 32:].permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=2e-6) and float(wide[..., :32].abs().max()) == 0
    # grid_sample gather (features NHWC + RGB NCHW), points inside, on and outside the border
    NC, P, C = 2, 500, 64
    img = torch.rand(NC, 3, 24, 24, generator=g)
    xy = torch.rand(NC, P, 2, generator=g) * 2.6 - 1.3
    xy[0, 0] = torch.tensor([1.0, -1.0]); xy[0, 1] = torch.tensor([-1.0, 1.0])
    rows = torch.full((NC * P, 80), 7.0, device=DEV)
    _run(_op(13, 1, p=(xd, img.to(DEV), xy.to(DEV), rows), i=(NC,) + _i64(P) + (17, 17, C, 24, 24, 80, 8)))
    fr = F.grid_sample(x, -xy.unsqueeze(2), align_corners=True, mode='bilinear', padding_mode='border')[..., 0].permute(0, 2, 1)
    rr = F.grid_sample(img, -xy.unsqueeze(2), align_corners=True, mode='bilinear', padding_mode='border')[..., 0].permute(0, 2, 1)
    got = rows.cpu().view(NC, P, 80)
    assert torch.allclose(got[..., 8:8 + C], fr, atol=3e-6) and torch.allclose(got[..., 8 + C:8 + C + 3], rr, atol=3e-6)
    assert (got[..., :8] == 7).all() and (got[..., 8 + C + 3:] == 7).all()
    # harmonic embedding with a broadcasting row map: out row r takes src row (r // 3) % 5
    src = torch.randn(5, 6, generator=g) * 3
    out = torch.zeros(15, 90, device=DEV)
    _run(_op(13, 2, p=(src.to(DEV), None, None, out), i=_i64(15) + (6, 90, 4) + _i64(3) + _i64(5) + _i64(1) + _i64(0)))
    ref = eft_ref.harmonic(src)[(torch.arange(15) // 3) % 5]
    assert torch.allclose(out.cpu()[:, 4:4 + 78], ref, atol=2e-6) and float(out[:, :4].abs().max()) == 0


@pytest.mark.parametrize("S,stride,gmul,groups", [(3, 40, 1, 40), (20, 1, 20, 6), (6, 7, 1, 7)])
def test_small_attention_and_softmax_pool(S, stride, gmul, groups):
    """Sequences are strided rows of one matrix: (S, stride, gmul) = (views, N*D, 1), (depths, 1, D), (views, N, 1)."""
    g = torch.Generator().manual_seed(S)
    M = S * groups
    qkv = torch.randn(M, 768, generator=g)
    out = torch.zeros(M, 256, device=DEV)
    _run(_op(13, 3, p=(qkv.to(DEV), None, None, out), i=_i64(groups) + (S,) + _i64(stride) + _i64(gmul), f=(1 / 16.0,)))
    rows = (torch.arange(groups)[:, None] * gmul + torch.arange(S)[None] * stride)          # [G, S]
    q, k, v = (qkv[rows][..., j * 256:(j + 1) * 256] for j in range(3))
    ref = torch.softmax(q @ k.transpose(1, 2) / 16.0, -1) @ v
    assert torch.allclose(out.cpu()[rows], ref, atol=1e-5)
    x, w, b = torch.randn(M, 256, generator=g), torch.randn(256, generator=g) * 0.2, torch.randn(1, generator=g)
    hw, hb = torch.randn(3, 256, generator=g) * 0.1, torch.randn(3, generator=g)
    pooled, rgb = torch.zeros(groups, 256, device=DEV), torch.zeros(groups, 3, device=DEV)
    _run(_op(13, 4, p=(x.to(DEV), w.to(DEV), b.to(DEV), pooled, hw.to(DEV), hb.to(DEV), rgb),
             i=_i64(groups) + (S,) + _i64(stride) + _i64(gmul)))
    xs = x[rows]
    p = torch.softmax(xs @ w + b, dim=1)
    ref = (xs * p[..., None]).sum(1)
    assert torch.allclose(pooled.cpu(), ref, atol=1e-5)
    assert torch.allclose(rgb.cpu(), torch.sigmoid(ref @ hw.T + hb), atol=1e-5)


def _module():
    from sparsefusion_amd.eft import EpipolarFeatureTransformer, eft_param_spec
    assert [(k, tuple(s)) for k, s in eft_param_spec(False)] == spec()                 # names, order and shapes of the reference
    net = EpipolarFeatureTransformer(use_r=True, encoder='resnet18', return_features=True, remove_unused_layers=False)
    r = net.load_state_dict(state(0), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    return net.to(DEV)


@pytest.mark.parametrize("name", ["small", "six_views"])
def test_eft_matches_reference_golden(name):
    import collections
    RayBundle = collections.namedtuple("RayBundle", ["origins", "directions", "lengths", "xys"])
    G = torch.load(f"{GOLD}/eft_forward.pt")[name]
    cams, images, o, d, lengths = scene(**G["cfg"])
    net = _module()
    rb = RayBundle(o.to(DEV), d.to(DEV), lengths.to(DEV), None)
    rgb, f3, zero = net(rb, input_cameras=cams.to(DEV), input_rgb=images.to(DEV))
    rgb, f3 = rgb.cpu(), f3.cpu()
    cos = F.cosine_similarity(f3.flatten().double(), G["f3"].flatten().double(), dim=0).item()
    print(f"{name}: f3 rel L2 {rel_err(f3, G['f3']):.3e} cos {cos:.6f}; rgb max abs {float((rgb - G['rgb']).abs().max()):.3e}")
    assert zero == 0 and rgb.shape == G["rgb"].shape and f3.shape == G["f3"].shape
    assert rel_err(f3, G["f3"]) < 3e-2 and cos > 0.999
    assert float((rgb - G["rgb"]).abs().max()) < 1e-2
    # encoder pyramid against the oracle's (localises errors): [NC, 512, R/2, R/2]
    with torch.no_grad():
        lat_ref = eft_ref.encode(state(0), images)
    lat = net.encoder_latent.cpu()
    assert lat.shape == lat_ref.shape and rel_err(lat, lat_ref) < 2e-2
    # the chunked entry point of the pre-pass gives the same rows (distillation.py:106 uses n_batches = 16)
    rb3 = RayBundle(o[None].to(DEV), d[None].to(DEV), lengths[None].to(DEV), None)
    # (r04: chunks are merged up to max_tokens_per_call tokens -- rays are independent, the chunking only bounds memory; both the
    # merged single plan and four real chunks are checked)
    rgb_b, f3_b, _ = net.batched_forward(rb3, n_batches=4)
    assert rgb_b.shape == (1, o.shape[0], 3) and f3_b.shape == (1, o.shape[0], 256)
    assert rel_err(f3_b[0].cpu(), f3) < 2e-2 and float((rgb_b[0].cpu() - rgb).abs().max()) < 5e-3
    n_fwd = [0]
    orig = net.forward
    net.forward = lambda *a, **k: (n_fwd.__setitem__(0, n_fwd[0] + 1), orig(*a, **k))[1]
    net.max_tokens_per_call = 1                                        # never merge: the reference's chunking
    rgb_c, f3_c, _ = net.batched_forward(rb3, n_batches=4)
    assert n_fwd[0] == 4
    assert rel_err(f3_c[0].cpu(), f3) < 2e-2 and float((rgb_c[0].cpu() - rgb).abs().max()) < 5e-3
    assert rel_err(f3_c[0].cpu(), f3_b[0].cpu()) < 1e-2          # other M -> other conv kernels / tiles: bf16-path noise (measured 3.9e-3)


def test_eft_transformer_linears_on_operand_twins_match_fp32_reads():
    """r04: the K = 256 linears of the three transformers read operand-type twins that their producers (LayerNorm, ReLU linear, the
    attention core) leave, through the LDS-DMA conv kernel; `linear_twin = False` plans the fp32 reads of r03.  Same operand values
    (the fp32 path rounds on load), other kernels: equal to accumulation order.  Needs >= 1024 tokens per transformer."""
    import collections
    RayBundle = collections.namedtuple("RayBundle", ["origins", "directions", "lengths", "xys"])
    cams, images, o, d, lengths = scene(NC=3, R=64, N=96, D=20, seed=7)
    outs = {}
    for on in (True, False):
        net = _module()
        net.linear_twin = on
        rb = RayBundle(o.to(DEV), d.to(DEV), lengths.to(DEV), None)
        rgb, f3, _ = net(rb, input_cameras=cams.to(DEV), input_rgb=images.to(DEV))
        outs[on] = (rgb.cpu(), f3.cpu())
        plan = [v for k, v in net._plans.items() if k[0] == "fwd"][0]
        from sparsefusion_amd.unet import OP_CONV
        n16 = sum(1 for op in plan.ops if op.type == OP_CONV and not (op.flags & 1))
        assert (n16 >= 24) if on else (n16 == 0), n16          # (the third transformer runs on N * NC = 288 tokens here: below the twin threshold)
    assert rel_err(outs[True][1], outs[False][1]) < 5e-3 and float((outs[True][0] - outs[False][0]).abs().max()) < 5e-3


def test_eft_rejects_unsupported():
    from sparsefusion_amd.eft import EpipolarFeatureTransformer
    with pytest.raises(NotImplementedError):
        EpipolarFeatureTransformer(encoder='lite', return_features=True)
    with pytest.raises(NotImplementedError):
        EpipolarFeatureTransformer(encoder='resnet18', return_features=False)
    net = EpipolarFeatureTransformer(encoder='resnet18', return_features=True)
    with pytest.raises(RuntimeError):
        net.encode(None, torch.zeros(2, 3, 64, 64))                  # CPU tensor: no fallback


def test_eft_feature_render_matches_reference_renderer():
    """Row E1, renderer half: `renderer_feat(cameras=, volumetric_function=eft.batched_forward, n_batches=16, input_cameras=,
    input_rgb=)` (sparsefusion/distillation.py:103-109) through the drop-ins of sparsefusion_amd/utils against the reference's
    own CustomImplicitRenderer + LightFieldRaymarcher + EFT run on CPU (tests/golden/make_golden_eft_render.py)."""
    from sparsefusion_amd.utils.cameras import PinholeCameras as Cams
    from sparsefusion_amd.utils.render_utils import init_light_field_renderer
    import math
    G = torch.load(f"{GOLD}/eft_render.pt")
    cfg = G["cfg"]
    cams, images, _, _, _ = scene(cfg["NC"], cfg["R"], 4, 20, cfg["seed"])
    net = _module()
    _, _, renderer_feat = init_light_field_renderer(0, cfg["R"], cfg["R"], min=cfg["min_depth"], max=cfg["max_depth"],
                                                    scale_factor=cfg["scale_factor"])
    a = 0.25
    c, s = math.cos(a), math.sin(a)
    q = Cams(torch.tensor([[[c, 0, -s], [0, 1, 0], [s, 0, c]]]), torch.tensor([[0.02, 0.03, 4.1]]), torch.full((1, 2), 2.2)).to(DEV)
    in_cams = Cams(cams.R, cams.T, cams.focal).to(DEV)
    net.encode(in_cams, images.to(DEV))
    feats, bundle, reg = renderer_feat(cameras=q, volumetric_function=net.batched_forward, n_batches=16, input_cameras=in_cams,
                                       input_rgb=images.to(DEV))
    assert reg == 0 and feats.shape == G["features"].shape == (1, 8, 8, 259)
    assert torch.allclose(bundle.origins.cpu(), G["origins"], atol=1e-5) and torch.allclose(bundle.directions.cpu(), G["directions"], atol=1e-5)
    rgb, f3 = feats.cpu().split([3, 256], dim=-1)
    rgb_g, f3_g = G["features"].split([3, 256], dim=-1)
    cos = F.cosine_similarity(f3.flatten().double(), f3_g.flatten().double(), dim=0).item()
    print(f"feature render: f3 rel L2 {rel_err(f3, f3_g):.3e} cos {cos:.6f}; rgb max abs {float((rgb - rgb_g).abs().max()):.3e}")
    assert rel_err(f3, f3_g) < 3e-2 and cos > 0.999 and float((rgb - rgb_g).abs().max()) < 1e-2
