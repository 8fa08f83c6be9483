"""Import the REAL reference (read-only, /root/reference) on CPU -- dev container only.

Used to (a) pin the restatements in this package against the reference's own Python and
(b) generate tests/golden/*.pt (tests/golden/make_golden.py).  /root/reference does not exist
on the GPU box, so nothing in the gpu tests, smoke() or bench.py calls into this module.

Recipe = SURVEY.md Appendix A: stub modules for the import-time-only dependencies, a stub
`raymarching` package and a stub `_gridencoder` extension backed by oracle/ngp_ref.c (the
reference has no CPU implementation of those two native entry points)."""
import argparse
import os
import sys
import types
import warnings

import torch

REFERENCE_ROOT = os.environ.get("SPARSEFUSION_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "external"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    warnings.filterwarnings("ignore", category=FutureWarning)
    warnings.filterwarnings("ignore", category=UserWarning)
    from . import ngp_native

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in ("torchvision", "torchvision.transforms", "cv2", "trimesh", "mcubes", "imageio", "tensorboardX",
                 "lpips"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    if "torch_ema" not in sys.modules:
        _stub("torch_ema", ExponentialMovingAverage=object)

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        o = rays_o.float().contiguous().view(-1, 3)
        d = rays_d.float().contiguous().view(-1, 3)
        nears, fars = torch.empty(o.shape[0]), torch.empty(o.shape[0])
        ngp_native.near_far_from_aabb(o, d, aabb.float().contiguous(), o.shape[0], min_near, nears, fars)
        return nears, fars

    _stub("raymarching", near_far_from_aabb=near_far_from_aabb, **_occupancy_api(ngp_native))
    _stub("_gridencoder", grid_encode_forward=ngp_native.grid_encode_forward,
          grid_encode_backward=ngp_native.grid_encode_backward)
    _installed = True


def _occupancy_api(nn):
    """CPU stand-ins for the Python-level functions of the reference's raymarching/raymarching.py:80-380 that its
    `run_cuda` / `update_extra_state` call, on top of the C oracle (the reference's own versions force .cuda()).
    Argument handling (M, align, noises, slicing to the counted points) restates raymarching.py:160-235, :296-346."""

    def morton3D(coords):
        c = coords.int().contiguous()
        out = torch.empty(c.shape[0], dtype=torch.int32)
        nn.morton3D(c, c.shape[0], out)
        return out

    def morton3D_invert(indices):
        i = indices.int().contiguous()
        out = torch.empty(i.shape[0], 3, dtype=torch.int32)
        nn.morton3D_invert(i, i.shape[0], out)
        return out

    def packbits(grid, thresh, bitfield=None):
        g = grid.contiguous()
        C_, H3 = g.shape
        n = C_ * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(n, dtype=torch.uint8)
        nn.packbits(g.view(-1), n, float(thresh), bitfield)
        return bitfield

    def march_rays_train(rays_o, rays_d, bound, density_bitfield, C_, H, nears, fars, step_counter=None, mean_count=-1,
                         perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        rays = torch.empty(N, 3, dtype=torch.int32)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32)
        noises = torch.rand(N) if perturb else torch.zeros(N)
        nn.march_rays_train(rays_o, rays_d, density_bitfield.contiguous(), float(bound), float(dt_gamma), max_steps, N, C_, H, M,
                            nears, fars, xyzs, dirs, deltas, rays, step_counter, noises)
        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays

    class _CompositeTrain(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
            sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
            M, N = sigmas.shape[0], rays.shape[0]
            weights_sum, depth, image = torch.empty(N), torch.empty(N), torch.empty(N, 3)
            nn.composite_rays_train_forward(sigmas, rgbs, deltas.contiguous(), rays, M, N, T_thresh, weights_sum, depth, image)
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
            ctx.dims = [M, N, T_thresh]
            return weights_sum, depth, image

        @staticmethod
        def backward(ctx, grad_weights_sum, grad_depth, grad_image):
            sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
            M, N, T_thresh = ctx.dims
            grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
            nn.composite_rays_train_backward(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs,
                                             deltas.contiguous(), rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs)
            return grad_sigmas, grad_rgbs, None, None, None

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C_, H, near, far, align=-1,
                   perturb=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        noises = torch.rand(n_alive) if perturb else torch.zeros(n_alive)
        nn.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, float(bound), float(dt_gamma), max_steps, C_, H,
                      density_bitfield.contiguous(), near, far, xyzs, dirs, deltas, noises)
        return xyzs, dirs, deltas

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        nn.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(), deltas,
                          weights_sum, depth, image)
        return tuple()

    return dict(morton3D=morton3D, morton3D_invert=morton3D_invert, packbits=packbits, march_rays_train=march_rays_train,
                composite_rays_train=_CompositeTrain.apply, march_rays=march_rays, composite_rays=composite_rays)


def ngp_opt():
    """The 21 fields of get_default_torch_ngp_opt (sparsefusion/distillation.py:500-525)."""
    opt = argparse.Namespace()
    opt.cuda_ray = False; opt.max_steps = 256; opt.num_steps = 64; opt.upsample_steps = 64
    opt.update_extra_interval = 16; opt.max_ray_batch = 4096; opt.albedo_iters = 1000; opt.bg_radius = 0
    opt.density_thresh = 10; opt.fp16 = True; opt.backbone = 'grid'; opt.w = 128; opt.h = 128; opt.hw_scale = 2
    opt.bound = 4; opt.min_near = 0.1; opt.dt_gamma = 0; opt.lambda_entropy = 1e-4; opt.lambda_opacity = 0
    opt.lambda_orient = 1e-2; opt.lambda_smooth = 0
    return opt


def reference_ngp(opt=None):
    """The reference's own NeRFNetwork (external/nerf/network_grid.py:36) on CPU."""
    install()
    from external.nerf.network_grid import NeRFNetwork
    return NeRFNetwork(opt or ngp_opt())


UNET_KWARGS = dict(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2),
                   layer_attns=(False, False, False, True), layer_cross_attns=(False, False, False, False),
                   cond_images_channels=256, attn_pool_text=False)   # utils/load_model.py:58-69


def reference_vldm(unet_kwargs=None):
    """Reference Unet + DDPM with the canonical hyper-parameters (utils/load_model.py:58-91)."""
    install()
    from external.imagen_pytorch import Unet
    from sparsefusion.vldm import DDPM
    unet = Unet(**(unet_kwargs or UNET_KWARGS))
    vldm = DDPM(channels=4, unets=(unet,), conditional_encoder=None, conditional_embed_dim=None,
                image_sizes=(32,), timesteps=500, cond_drop_prob=0.1, pred_objectives='noise', conditional=False,
                auto_normalize_img=False, clip_output=True, dynamic_thresholding=False,
                dynamic_thresholding_percentile=.68, clip_value=10)
    return vldm


def reference_plms(vldm, steps=50):
    install()
    from external.plms import PLMSSampler
    return PLMSSampler(vldm, steps)


def reference_vae(cfg):
    """The reference's own Encoder / Decoder (external/ldm/modules/diffusionmodules/model.py:368-569) wired as
    AutoencoderKL.__init__ does (external/ldm/models/autoencoder.py:296-303).  AutoencoderKL itself cannot be
    imported here (pytorch_lightning, taming); its encode()/decode() are three lines each (:324-333) and are
    replayed by `encode_mode` / `decode` below with the reference's DiagonalGaussianDistribution."""
    install()
    from external.ldm.modules.diffusionmodules.model import Encoder, Decoder
    dd = {k: v for k, v in cfg.items() if k != "embed_dim"}
    dd["attn_resolutions"] = list(dd["attn_resolutions"])
    m = torch.nn.Module()
    m.encoder = Encoder(**dd)
    m.decoder = Decoder(**dd)
    m.quant_conv = torch.nn.Conv2d(2 * cfg["z_channels"], 2 * cfg["embed_dim"], 1)
    m.post_quant_conv = torch.nn.Conv2d(cfg["embed_dim"], cfg["z_channels"], 1)

    def encode_mode(x):
        from external.ldm.modules.distributions.distributions import DiagonalGaussianDistribution
        return DiagonalGaussianDistribution(m.quant_conv(m.encoder(x))).mode()

    def decode(z):
        return m.decoder(m.post_quant_conv(z))

    m.encode_mode, m.decode = encode_mode, decode
    return m


# ---------------------------------------------------------------------------------------------
# EFT (row E1): the reference module itself imports on CPU once its third-party imports are stubbed
# ---------------------------------------------------------------------------------------------
class PinholeCameras:
    """Minimal stand-in for pytorch3d's PerspectiveCameras: the two methods sparsefusion/eft.py calls
    (`transform_points_ndc` :239, `get_camera_center` :316) and `__len__`.  World -> view is X_cam = X_world R + T
    (row vectors, the pytorch3d convention), NDC = focal * (x, y) / z + principal point.  pytorch3d itself is
    absent and unpinned (ENVIRONMENT.md:42): the tests use this class on BOTH sides, so the convention cancels."""

    def __init__(self, R, T, focal, principal=None):
        self.R, self.T = R.float(), T.float()                       # [NC,3,3], [NC,3]
        self.focal = focal.float()                                  # [NC,2]
        self.principal = torch.zeros_like(self.focal) if principal is None else principal.float()

    def __len__(self):
        return self.R.shape[0]

    def to(self, device):
        return PinholeCameras(self.R.to(device), self.T.to(device), self.focal.to(device), self.principal.to(device))

    def get_camera_center(self):
        return -torch.bmm(self.T[:, None], self.R.transpose(1, 2))[:, 0]          # C = -T R^T

    def transform_points_ndc(self, pts):
        p = pts.expand(len(self), -1, -1) if pts.shape[0] == 1 else pts
        cam = torch.bmm(p, self.R) + self.T[:, None]
        z = cam[..., 2:3]
        xy = cam[..., :2] / z * self.focal[:, None] + self.principal[:, None]
        return torch.cat([xy, 1.0 / z], -1)


def _resnet18_stub():
    """torchvision.models.resnet18 (torchvision 0.12, ENVIRONMENT.md) restated with torch.nn so that the reference's
    `getattr(torchvision.models, 'resnet18')(pretrained=True)` (eft.py:99) constructs; weights are random here."""
    import torch.nn as nn

    class BasicBlock(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idt)

    class ResNet18(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
            self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
            self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
            self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512, 1))
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512, 1000)

    return lambda pretrained=False, **kw: ResNet18()


def reference_eft(**kwargs):
    """The reference's EpipolarFeatureTransformer (sparsefusion/eft.py:55) with the arguments of utils/load_model.py:33."""
    install()
    import collections
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "models"):
        tv.models = _stub("torchvision.models", resnet18=_resnet18_stub())
    RayBundle = collections.namedtuple("RayBundle", ["origins", "directions", "lengths", "xys"])

    def ray_bundle_to_ray_points(rb):
        return rb.origins[..., None, :] + rb.lengths[..., :, None] * rb.directions[..., None, :]

    for name in ("pytorch3d", "pytorch3d.renderer", "pytorch3d.renderer.cameras", "pytorch3d.renderer.implicit",
                 "pytorch3d.renderer.implicit.utils", "skimage", "skimage.metrics"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["pytorch3d.renderer"].RayBundle = RayBundle
    sys.modules["pytorch3d.renderer"].ray_bundle_to_ray_points = ray_bundle_to_ray_points
    sys.modules["pytorch3d.renderer.cameras"].PerspectiveCameras = PinholeCameras
    sys.modules["pytorch3d.renderer.implicit.utils"]._validate_ray_bundle_variables = lambda *a, **k: None
    sys.modules["pytorch3d.renderer.implicit.utils"].ray_bundle_variables_to_ray_points = \
        lambda o, d, l: o[..., None, :] + l[..., :, None] * d[..., None, :]
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    from sparsefusion.eft import EpipolarFeatureTransformer
    kw = dict(use_r=True, encoder='resnet18', return_features=True, remove_unused_layers=False)
    kw.update(kwargs)
    return EpipolarFeatureTransformer(**kw), RayBundle


# ---------------------------------------------------------------------------------------------
# EFT feature renderer (row E1, second half): the reference's CustomImplicitRenderer / LightFieldRaymarcher import on CPU
# once pytorch3d's names exist; the ray sampler is pytorch3d's own class (absent, unpinned) and is restated below.
# ---------------------------------------------------------------------------------------------
def _unproject(cams, xy_depth):
    """PerspectiveCameras.unproject_points (NDC): inverse of PinholeCameras.transform_points_ndc at the given depth."""
    xy, depth = xy_depth[..., :2], xy_depth[..., 2:3]
    cam = torch.cat([(xy - cams.principal[:, None]) / cams.focal[:, None] * depth, depth], -1)
    return torch.bmm(cam - cams.T[:, None], cams.R.transpose(1, 2))


class GridRaysamplerRef:
    """Restatement of pytorch3d's GridRaysampler + _xy_to_ray_bundle (published algorithm): lattice of
    NDC points y-major from (min_x, min_y) to (max_x, max_y) inclusive, ray = un-projection at depths 1 and 2,
    directions = p2 - p1 (not normalised), origins = p1 - directions, lengths = linspace(min_depth, max_depth, n)."""

    def __init__(self, min_x, max_x, min_y, max_y, image_width, image_height, n_pts_per_ray, min_depth, max_depth):
        self.args = (min_x, max_x, min_y, max_y, image_width, image_height, n_pts_per_ray, min_depth, max_depth)

    def __call__(self, cameras, **kwargs):
        import collections
        min_x, max_x, min_y, max_y, W, H, n, dmin, dmax = self.args
        RayBundle = collections.namedtuple("RayBundle", ["origins", "directions", "lengths", "xys"])
        N = len(cameras)
        xy = torch.zeros(N, H, W, 2)
        for r in range(H):
            for c in range(W):
                xy[:, r, c, 0] = min_x + (max_x - min_x) * c / (W - 1)
                xy[:, r, c, 1] = min_y + (max_y - min_y) * r / (H - 1)
        flat = xy.view(N, H * W, 2)
        p1 = _unproject(cameras, torch.cat([flat, torch.ones(N, H * W, 1)], -1))
        p2 = _unproject(cameras, torch.cat([flat, 2 * torch.ones(N, H * W, 1)], -1))
        d = p2 - p1
        lengths = torch.linspace(dmin, dmax, n)[None, None].expand(N, H * W, n)
        return RayBundle((p1 - d).view(N, H, W, 3), d.view(N, H, W, 3), lengths.reshape(N, H, W, n), xy)


def reference_eft_renderer(raysampler):
    """The reference's `CustomImplicitRenderer(raysampler, LightFieldRaymarcher(), reg=True)` (utils/render_utils.py:170-183)."""
    install()
    for name in ("pytorch3d", "pytorch3d.ops", "pytorch3d.ops.utils", "pytorch3d.structures", "pytorch3d.transforms", "pytorch3d.renderer",
                 "pytorch3d.renderer.cameras", "pytorch3d.renderer.implicit", "pytorch3d.renderer.implicit.raysampling",
                 "pytorch3d.renderer.implicit.utils", "pytorch3d.renderer.implicit.raymarching"):
        if name not in sys.modules:
            _stub(name)
    m = sys.modules
    m["pytorch3d.ops.utils"].eyes = None
    m["pytorch3d.structures"].Volumes = None
    m["pytorch3d.transforms"].Transform3d = None
    m["pytorch3d.renderer.cameras"].CamerasBase = object
    m["pytorch3d.renderer.implicit.raysampling"].RayBundle = object
    for n in ("_validate_ray_bundle_variables", "ray_bundle_variables_to_ray_points", "ray_bundle_to_ray_points"):
        if not hasattr(m["pytorch3d.renderer.implicit.utils"], n):
            setattr(m["pytorch3d.renderer.implicit.utils"], n, None)
    m["pytorch3d.renderer"].EmissionAbsorptionRaymarcher = object
    for n in ("_check_density_bounds", "_check_raymarcher_inputs", "_shifted_cumprod"):
        setattr(m["pytorch3d.renderer.implicit.raymarching"], n, None)
    from utils.eft_renderer import CustomImplicitRenderer
    from utils.eft_raymarcher import LightFieldRaymarcher
    return CustomImplicitRenderer(raysampler=raysampler, raymarcher=LightFieldRaymarcher(), reg=True)
