"""torch-CPU fp32 restatement of LPIPS(net='vgg') as the reference uses it -- TEST INFRASTRUCTURE ONLY.

Call site: external/external_utils.py:11-50 (`PerceptualLoss('vgg')`, `normalize=True`: inputs in [0,1] are mapped to
[-1,1]) <- sparsefusion/distillation.py:161,312-314.  The algorithm itself lives in the third-party package `lpips`
(requirements.txt:15, UNPINNED; latest published 0.1.4), which is NOT installed here and not vendored under
/root/reference, and it needs torchvision's VGG16 (absent too).  What follows restates the published algorithm
(Zhang et al., CVPR 2018; lpips/lpips.py `LPIPS.forward`, `ScalingLayer`, `NetLinLayer`, `normalize_tensor`,
`spatial_average`; lpips/pretrained_networks.py `vgg16` slices):
    in -> (in - shift) / scale -> VGG16 conv stack -> features after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
    per layer: unit-normalise channels (x / (|x|_2 + 1e-10)), squared difference, 1x1 conv with non-negative weights
    (no bias), spatial mean; sum over the 5 layers -> [B, 1, 1, 1]
PARITY UNPINNED: no golden vector can be produced without the package / pretrained weights.  State-dict key names
follow lpips 0.1.4 as published (`net.slice{k}.{idx}.weight`, `lin{k}.model.1.weight`, `scaling_layer.shift/scale`)."""
import math

import torch
import torch.nn.functional as F

SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)
# (slice, torchvision vgg16.features index, Cin, Cout); a 2x2 max-pool opens slices 2..5
VGG_CONVS = [(1, 0, 3, 64), (1, 2, 64, 64),
             (2, 5, 64, 128), (2, 7, 128, 128),
             (3, 10, 128, 256), (3, 12, 256, 256), (3, 14, 256, 256),
             (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512),
             (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]
CHNS = (64, 128, 256, 512, 512)


def lpips_param_spec():
    spec = []
    for sl, idx, cin, cout in VGG_CONVS:
        spec += [(f"net.slice{sl}.{idx}.weight", (cout, cin, 3, 3)), (f"net.slice{sl}.{idx}.bias", (cout,))]
    for k, c in enumerate(CHNS):
        spec.append((f"lin{k}.model.1.weight", (1, c, 1, 1)))
    return spec


def init_state(seed=0):
    """Synthetic stand-in for the pretrained weights: variance-preserving (He) convs so that activations stay O(1)
    through 13 ReLU layers, small biases, non-negative lin weights (the published ones are clamped >= 0)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in lpips_param_spec():
        if name.startswith("lin"):
            t = torch.rand(shape, generator=g) / shape[1] * 4
        elif name.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * 9
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        sd[name] = t
    return sd


class _ConvRounded(torch.autograd.Function):
    """3x3 conv whose MATRIX OPERANDS are rounded to `dt` on both passes, fp32 accumulation: forward conv(round(x), round(w)) + b,
    backward-data conv_transpose(round(grad), round(w)) -- what an MFMA path with `dt` operands computes (no weight gradient: the
    VGG is frozen).  Used to attribute the HIP path's gradient error to operand rounding (tests/test_gpu_lpips.py)."""

    @staticmethod
    def forward(ctx, x, w, b, dt):
        r = lambda t: t.to(dt).float()
        ctx.save_for_backward(w)
        ctx.dt = dt
        return F.conv2d(r(x), r(w), b, padding=1)

    @staticmethod
    def backward(ctx, gy):
        (w,) = ctx.saved_tensors
        r = lambda t: t.to(ctx.dt).float()
        return F.conv_transpose2d(r(gy), r(w), padding=1), None, None, None


def vgg_features(sd, x, operand_dtype=None):
    feats, cur = [], 1
    h = x
    for sl, idx, cin, cout in VGG_CONVS:
        if sl != cur:
            feats.append(h)
            h = F.max_pool2d(h, kernel_size=2, stride=2)
            cur = sl
        w, b = sd[f"net.slice{sl}.{idx}.weight"], sd[f"net.slice{sl}.{idx}.bias"]
        h = F.relu(F.conv2d(h, w, b, padding=1) if operand_dtype is None else _ConvRounded.apply(h, w, b, operand_dtype))
    feats.append(h)
    return feats


def _normalize(x, eps=1e-10):
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def lpips(sd, in0, in1, normalize=True, operand_dtype=None, heads=range(5)):
    """in0, in1: [B, 3, H, W]; normalize=True expects [0, 1] inputs (external_utils.py:37-39). Returns [B, 1, 1, 1].
    `operand_dtype` (e.g. torch.bfloat16): emulate rounded conv operands (see _ConvRounded); `heads`: the feature taps summed
    (all five in LPIPS; a subset localises an error to a VGG slice)."""
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    shift = torch.tensor(SHIFT).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE).view(1, 3, 1, 1)
    f0 = vgg_features(sd, (in0 - shift) / scale, operand_dtype)
    f1 = vgg_features(sd, (in1 - shift) / scale, operand_dtype)
    val = 0
    for k in heads:
        diff = (_normalize(f0[k]) - _normalize(f1[k])) ** 2
        val = val + F.conv2d(diff, sd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return val
