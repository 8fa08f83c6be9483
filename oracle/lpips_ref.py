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


def vgg_features(sd, x):
    feats, cur = [], 1
    h = x
    for sl, idx, cin, cout in VGG_CONVS:
        if sl != cur:
            feats.append(h)
            h = F.max_pool2d(h, kernel_size=2, stride=2)
            cur = sl
        h = F.relu(F.conv2d(h, sd[f"net.slice{sl}.{idx}.weight"], sd[f"net.slice{sl}.{idx}.bias"], padding=1))
    feats.append(h)
    return feats


def _normalize(x, eps=1e-10):
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def lpips(sd, in0, in1, normalize=True):
    """in0, in1: [B, 3, H, W]; normalize=True expects [0, 1] inputs (external_utils.py:37-39). Returns [B, 1, 1, 1]."""
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    shift = torch.tensor(SHIFT).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE).view(1, 3, 1, 1)
    f0 = vgg_features(sd, (in0 - shift) / scale)
    f1 = vgg_features(sd, (in1 - shift) / scale)
    val = 0
    for k in range(5):
        diff = (_normalize(f0[k]) - _normalize(f1[k])) ** 2
        val = val + F.conv2d(diff, sd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return val
