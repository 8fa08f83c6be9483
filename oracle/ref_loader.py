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

    _stub("raymarching", near_far_from_aabb=near_far_from_aabb)
    _stub("_gridencoder", grid_encode_forward=ngp_native.grid_encode_forward,
          grid_encode_backward=ngp_native.grid_encode_backward)
    _installed = True


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
