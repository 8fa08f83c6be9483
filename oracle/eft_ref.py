"""torch-CPU fp32 restatement of the Epipolar Feature Transformer forward (row E1) -- TEST INFRASTRUCTURE ONLY.

Functional over a plain state dict with the reference's 278 keys; each piece cites what it follows in /root/reference:
  encode            sparsefusion/eft.py:155-207   (resnet18 trunk conv1..layer3, bilinear align_corners resize, concat: 512 ch)
  plucker / harmonic  :209-215, utils/common_utils.py:68-155 (sin | cos | x, 6 octaves, dim-major)
  index             :217-336  (project to the input views, grid_sample(bilinear, border, align_corners) of features and RGB,
                               reference Pluecker rays, depth embedding)
  forward           :351-452  (T1 over views, T2 over depths + softmax pooling, T3 over views + softmax pooling, colour head)
  TransformerEncoder  :19-52  (Linear+GELU, 4 x nn.TransformerEncoderLayer(256, nhead 1, ff 256, post-norm, ReLU), seq-first)
  resnet18          torchvision 0.12 `models/resnet.py` (BasicBlock; third-party, absent: architecture restated, weights synthetic)
Cameras are duck-typed: `transform_points_ndc(xyz) -> [NC, P, 3]` and `get_camera_center() -> [NC, 3]` (pytorch3d's
PerspectiveCameras in the reference; absent and unpinned, so the tests use oracle.ref_loader.PinholeCameras on both sides).
Pinned against the REAL reference module (imported on CPU by oracle/ref_loader.reference_eft) through tests/golden/eft_forward.pt."""
import math

import torch
import torch.nn.functional as F


def init_state(spec, seed=0):
    """Deterministic synthetic weights for the (name, shape) spec: He convs, near-identity BatchNorm with non-trivial
    running statistics, variance-preserving linears, small biases."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in spec:
        if name.endswith("num_batches_tracked"):
            t = torch.tensor(100)
        elif name.endswith("running_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif ".bn" in name or "downsample.1" in name or ".norm" in name:
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 4:
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[-1])
        sd[name] = t
    return sd


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _basic_block(sd, p, x, stride):
    idt = x
    if (p + ".downsample.0.weight") in sd:
        idt = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1))
    return F.relu(out + idt)


def encode(sd, images):
    """eft.py:173-206 with encoder_num_layers = 4: [NC,3,H,W] -> [NC,512,H/2,W/2]."""
    e = "encoder_model"
    x = F.relu(_bn(sd, e + ".bn1", F.conv2d(images, sd[e + ".conv1.weight"], sd.get(e + ".conv1.bias"), stride=2, padding=3)))
    latents = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    for layer, stride in ((1, 1), (2, 2), (3, 2)):
        x = _basic_block(sd, f"{e}.layer{layer}.0", x, stride)
        x = _basic_block(sd, f"{e}.layer{layer}.1", x, 1)
        latents.append(x)
    sz = latents[0].shape[-2:]
    return torch.cat([F.interpolate(l, sz, mode='bilinear', align_corners=True) for l in latents], dim=1)


def harmonic(x, n=6, omega0=1.0):
    freqs = (2.0 ** torch.arange(n, dtype=torch.float32)) * omega0
    embed = (x[..., None] * freqs).view(*x.shape[:-1], -1)
    return torch.cat((embed.sin(), embed.cos(), x), dim=-1)


def plucker(origins, dirs):
    return harmonic(torch.cat((dirs, torch.cross(origins, dirs, dim=-1)), dim=-1))


def _encoder_layer(sd, p, x):
    """nn.TransformerEncoderLayer(256, 1, 256, dropout off): x [S, B, E], post-norm, ReLU."""
    S, B, E = x.shape
    qkv = F.linear(x, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    att = torch.softmax(torch.einsum("sbe,tbe->bst", q, k) / math.sqrt(E), dim=-1)
    sa = torch.einsum("bst,tbe->sbe", att, v)
    sa = F.linear(sa, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
    x = F.layer_norm(x + sa, (E,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    ff = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"],
                  sd[p + ".linear2.bias"])
    return F.layer_norm(x + ff, (E,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)


def transformer(sd, p, w):
    out = F.gelu(F.linear(w, sd[p + ".pre.0.weight"], sd[p + ".pre.0.bias"]))
    for i in range(4):
        out = _encoder_layer(sd, f"{p}.encoder.layers.{i}", out)
    return out


def eft_forward(sd, cameras, images, origins, directions, lengths, latent=None):
    """eft.py:351-452 with return_features=True, use_r=True: -> (rgb [N,3], f3 [N,256]).
    origins/directions [N,3], lengths [N,D]; images [NC,3,H,W] in the range the encoder was trained on."""
    if latent is None:
        latent = encode(sd, images)
    NC, N, D = images.shape[0], origins.shape[0], lengths.shape[1]
    xyz = origins[:, None, :] + lengths[:, :, None] * directions[:, None, :]                     # ray_bundle_to_ray_points
    dirs = F.normalize(directions, dim=-1)
    query = plucker(origins, dirs)[:, None, :]                                                   # [N,1,78]
    xy = cameras.transform_points_ndc(xyz.reshape(1, -1, 3))[..., :2].unsqueeze(2)               # [NC, N*D, 1, 2]
    feats = F.grid_sample(latent, -xy, align_corners=True, mode='bilinear', padding_mode='border')[..., 0].permute(0, 2, 1)
    rgbs = F.grid_sample(images, -xy, align_corners=True, mode='bilinear', padding_mode='border')[..., 0].permute(0, 2, 1)
    features = torch.cat((feats.reshape(NC, N, D, -1), rgbs.reshape(NC, N, D, -1)), dim=-1)      # [NC,N,D,515]
    centers = cameras.get_camera_center()[:, None, None, :].expand(NC, N, D, 3)
    in_dirs = F.normalize(xyz[None] - centers, dim=-1)
    ref = plucker(centers, in_dirs)                                                              # [NC,N,D,78]
    depths = harmonic(lengths[..., None])[None]                                                  # [1,N,D,13]
    # T1: sequence = views
    t1_in = torch.cat((ref.reshape(NC, N * D, -1), depths.expand(NC, -1, -1, -1).reshape(NC, N * D, -1),
                       features.reshape(NC, N * D, -1)), dim=-1)
    f1 = transformer(sd, "t1", t1_in).reshape(NC, N, D, -1)
    # T2: sequence = depths
    def dseq(t):                                                                                  # 'nc n d f -> d (nc n) f'
        return t.permute(2, 0, 1, 3).reshape(D, NC * N, -1)
    t2_in = torch.cat((dseq(query[None].expand(NC, -1, D, -1)), dseq(ref), dseq(depths.expand(NC, -1, -1, -1)), dseq(f1)), dim=-1)
    f2 = transformer(sd, "t2", t2_in).reshape(D, NC, N, -1).permute(1, 2, 0, 3)                   # [NC,N,D,F]
    w2 = torch.softmax(F.linear(f2, sd["t2_attn.weight"], sd["t2_attn.bias"]), dim=-2)
    f2 = (f2 * w2).sum(dim=-2)                                                                    # [NC,N,F]
    # T3: sequence = views
    t3_in = torch.cat((query.expand(-1, 1, -1)[None].expand(NC, -1, -1, -1)[..., 0, :], ref[..., D // 2, :], f2), dim=-1)
    f3 = transformer(sd, "t3", t3_in)
    w3 = torch.softmax(F.linear(f3, sd["t3_attn.weight"], sd["t3_attn.bias"]), dim=0)
    f3 = (f3 * w3).sum(dim=0)                                                                     # [N,F]
    rgb = torch.sigmoid(F.linear(f3, sd["color_layer.0.weight"], sd["color_layer.0.bias"]))
    return rgb, f3
