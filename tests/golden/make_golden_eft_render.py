"""Golden vectors of the EFT feature RENDER (row E1, renderer half) from the REAL reference classes (dev container only):
`CustomImplicitRenderer(raysampler, LightFieldRaymarcher(), reg=True)(cameras=, volumetric_function=eft.batched_forward,
n_batches=16, input_cameras=, input_rgb=)` as sparsefusion/distillation.py:103-109 calls it.  pytorch3d's GridRaysampler is
absent (unpinned third party): oracle.ref_loader.GridRaysamplerRef restates it and feeds the reference renderer."""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import eft_ref, ref_loader  # noqa: E402
import eft_common  # noqa: E402

CFG = dict(NC=3, R=64, scale_factor=8.0, min_depth=1.8, max_depth=4.2, seed=3)


def query_camera():
    a = 0.25
    c, s = math.cos(a), math.sin(a)
    R = torch.tensor([[[c, 0, -s], [0, 1, 0], [s, 0, c]]], dtype=torch.float32)
    return ref_loader.PinholeCameras(R, torch.tensor([[0.02, 0.03, 4.1]]), torch.full((1, 2), 2.2))


def sampler_args(R, scale_factor, min_depth, max_depth):
    hw = 1.0 / R
    n = int(R // scale_factor)
    return dict(min_x=1.0 - hw, max_x=-1.0 + hw, min_y=1.0 - hw, max_y=-1.0 + hw, image_width=n, image_height=n, n_pts_per_ray=20,
                min_depth=min_depth, max_depth=max_depth)


def main():
    eft, _ = ref_loader.reference_eft()
    eft.eval()
    eft.load_state_dict(eft_ref.init_state(eft_common.spec(), seed=0), strict=True)
    cams, images, _, _, _ = eft_common.scene(CFG["NC"], CFG["R"], 4, 20, CFG["seed"])
    sampler = ref_loader.GridRaysamplerRef(**sampler_args(CFG["R"], CFG["scale_factor"], CFG["min_depth"], CFG["max_depth"]))
    renderer = ref_loader.reference_eft_renderer(sampler)
    q = query_camera()
    eft.encode(cams, images)
    with torch.no_grad():
        feats, bundle, reg = renderer(cameras=q, volumetric_function=eft.batched_forward, n_batches=16, input_cameras=cams,
                                      input_rgb=images)
    print("features", tuple(feats.shape), "std %.4f" % feats.std().item(), "reg", reg)
    torch.save(dict(cfg=CFG, features=feats.clone(), origins=bundle.origins.clone(), directions=bundle.directions.clone(),
                    lengths=bundle.lengths.clone()), os.path.join(HERE, "eft_render.pt"))


if __name__ == "__main__":
    main()
