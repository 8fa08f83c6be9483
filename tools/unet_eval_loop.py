"""Run N plain UNet evals (canonical config, B from argv) -- target for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd.unet import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
unet.use_hip_graph = False
x, ls, cond = torch.randn(B, 4, 32, 32, device=dev), torch.zeros(B, device=dev), torch.randn(B, 256, 32, 32, device=dev)
if len(sys.argv) > 3 and sys.argv[3] == "full":      # the stand-alone forward (time path inside every eval)
    for _ in range(N):
        unet.forward(x, ls, cond_images=cond)
else:                                                 # what the PLMS sampler runs: time table once, plan body per eval
    ctx = unet.begin_sampling(cond, torch.linspace(-3, 3, 8, device=dev))
    for k in range(N):
        unet.eval_prepared(ctx, x, k % 8)
torch.cuda.synchronize()
