"""UNet / PLMS golden vectors from the REAL reference (dev container only; see make_golden.py)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader, unet_ref  # noqa: E402

CONFIGS = {"canonical": unet_ref.CANONICAL, "small": unet_ref.SMALL}


def _reference_unet(cfg):
    ref_loader.install()
    from external.imagen_pytorch import Unet
    return Unet(**cfg, layer_cross_attns=(False,) * 4, attn_pool_text=False, cond_on_z=False)


def inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["channels"], 32, 32, generator=g)
    cond = torch.randn(B, cfg["cond_images_channels"], 32, 32, generator=g)
    t = torch.tensor([0.37, 0.05, 0.9, 0.6][:B])
    return x, unet_ref.log_snr(t), cond


def make_unet():
    out = {}
    for name, cfg in CONFIGS.items():
        net = _reference_unet(cfg).eval()
        spec = [(k, list(v.shape)) for k, v in net.state_dict().items()]
        json.dump(spec, open(os.path.join(HERE, f"unet_keys_{name}.json"), "w"))
        sd = unet_ref.init_state([(k, tuple(s)) for k, s in spec], seed=0)
        net.load_state_dict(sd, strict=True)
        B = 2
        x, ls, cond = inputs(cfg, B, seed=5)
        with torch.no_grad():
            y = net.forward_with_cond_scale(x, ls, cond_images=cond, cond_scale=1.)
        out[name] = dict(B=B, input_seed=5, state_seed=0, y=y.clone())
        print(name, "params %.2fM" % (sum(v.numel() for v in sd.values()) / 1e6), "out std %.4f" % y.std().item())
    torch.save(out, os.path.join(HERE, "unet_forward.pt"))


def make_plms():
    """Reference PLMSSampler.sample on the small UNet inside the reference DDPM, seeded noise."""
    cfg = CONFIGS["small"]
    vldm = ref_loader.reference_vldm(dict(cfg, layer_cross_attns=(False,) * 4, attn_pool_text=False)).eval()
    unet = vldm.unets[0]
    spec = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    unet.load_state_dict(unet_ref.init_state(spec, seed=0), strict=True)
    sampler = ref_loader.reference_plms(vldm, 50)
    out = {}
    for max_thres in (0.005, 0.06, 0.995):
        B = 2
        g = torch.Generator().manual_seed(11)
        lat = 0.5 * torch.randn(B, 4, 32, 32, generator=g)
        cond = torch.randn(B, cfg["cond_images_channels"], 32, 32, generator=g)
        torch.manual_seed(77)
        with torch.no_grad():
            img, x_noisy, noise, acp = sampler.sample(lat.clone(), cond_images=cond, use_tqdm=False, return_noise=True,
                                                      max_thres=max_thres)
        out[max_thres] = dict(B=B, input_seed=11, noise_seed=77, img=img.clone(), x_noisy=x_noisy.clone(), noise=noise.clone(),
                              alpha_cumprod=acp.clone())
        print("plms", max_thres, img.std().item(), acp)
    torch.save(out, os.path.join(HERE, "plms_sample.pt"))


def make_plms_canonical():
    """Reference PLMSSampler.sample on the CANONICAL 400.68 M-parameter UNet at the distillation setting (max_thres = 0.5:
    50 steps = 51 evals, B = 1, sparsefusion/distillation.py:304) -- the headline configuration of BASELINE.json."""
    cfg = CONFIGS["canonical"]
    vldm = ref_loader.reference_vldm(dict(cfg, layer_cross_attns=(False,) * 4, attn_pool_text=False)).eval()
    unet = vldm.unets[0]
    spec = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    unet.load_state_dict(unet_ref.init_state(spec, seed=0), strict=True)
    sampler = ref_loader.reference_plms(vldm, 50)
    g = torch.Generator().manual_seed(13)
    lat = 0.5 * torch.randn(1, 4, 32, 32, generator=g)
    cond = torch.randn(1, cfg["cond_images_channels"], 32, 32, generator=g)
    torch.manual_seed(79)
    with torch.no_grad():
        img, x_noisy, noise, acp = sampler.sample(lat.clone(), cond_images=cond, use_tqdm=False, return_noise=True, max_thres=0.5)
    out = dict(B=1, input_seed=13, noise_seed=79, max_thres=0.5, img=img.clone(), x_noisy=x_noisy.clone(), noise=noise.clone(),
               alpha_cumprod=acp.clone())
    print("plms canonical", img.std().item(), acp)
    torch.save(out, os.path.join(HERE, "plms_sample_canonical.pt"))


def make_medium():
    """The dim-128 configuration (oracle/unet_ref.MEDIUM: every block on the fused kernels of the canonical plan): one forward of the
    reference Unet and one reference PLMSSampler.sample at max_thres = 0.5 (51 evals, B = 2) -> unet_medium.pt, unet_keys_medium.json."""
    cfg = unet_ref.MEDIUM
    net = _reference_unet(cfg).eval()
    spec = [(k, list(v.shape)) for k, v in net.state_dict().items()]
    json.dump(spec, open(os.path.join(HERE, "unet_keys_medium.json"), "w"))
    sd = unet_ref.init_state([(k, tuple(s)) for k, s in spec], seed=0)
    net.load_state_dict(sd, strict=True)
    x, ls, cond = inputs(cfg, 2, seed=5)
    with torch.no_grad():
        y = net.forward_with_cond_scale(x, ls, cond_images=cond, cond_scale=1.)
    out = {"forward": dict(B=2, input_seed=5, state_seed=0, y=y.clone())}
    vldm = ref_loader.reference_vldm(dict(cfg, layer_cross_attns=(False,) * 4, attn_pool_text=False)).eval()
    unet = vldm.unets[0]
    unet.load_state_dict(sd, strict=True)
    sampler = ref_loader.reference_plms(vldm, 50)
    g = torch.Generator().manual_seed(17)
    lat = 0.5 * torch.randn(2, 4, 32, 32, generator=g)
    cond = torch.randn(2, cfg["cond_images_channels"], 32, 32, generator=g)
    torch.manual_seed(81)
    with torch.no_grad():
        img, x_noisy, noise, acp = sampler.sample(lat.clone(), cond_images=cond, use_tqdm=False, return_noise=True, max_thres=0.5)
    out["plms"] = dict(B=2, input_seed=17, noise_seed=81, max_thres=0.5, img=img.clone(), x_noisy=x_noisy.clone(), noise=noise.clone(),
                       alpha_cumprod=acp.clone())
    print("medium params %.2fM" % (sum(v.numel() for v in sd.values()) / 1e6), "out std %.4f" % y.std().item(), "plms std", img.std().item())
    torch.save(out, os.path.join(HERE, "unet_medium.pt"))


if __name__ == "__main__":
    if "--medium" in sys.argv:
        make_medium()
    elif "--canonical-plms" in sys.argv:
        make_plms_canonical()
    else:
        make_unet()
        make_plms()
