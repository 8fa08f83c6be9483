"""Time the canonical UNet eval (graph replay and plain plan replay); prints launches per eval."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefusion_amd.unet import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
unet = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
            layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
for kv in [a for a in os.environ.get("SF_UNET_ATTRS", "").split(",") if a]:      # planner switches for A/B runs: "attn_in_out_proj=0,producer_slots=0"
    setattr(unet, kv.split("=")[0], int(kv.split("=")[1]))
for k, v in os.environ.items():
    if k == "SF_WAVES":
        unet.conv_waves_target = int(v)
    if k == "SF_LAZY":
        unet.lazy_consumers = int(v)
x, ls, cond = torch.randn(B, 4, 32, 32, device=dev), torch.zeros(B, device=dev), torch.randn(B, 256, 32, 32, device=dev)
for graph in (True,):
    unet.use_hip_graph = graph
    unet.invalidate()
    for _ in range(5):
        unet.forward(x, ls, cond_images=cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        unet.forward(x, ls, cond_images=cond)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    plan = unet._plan(B, dev)
    print(f"B={B} graph={graph} lazy={unet.lazy_consumers} ops={len(plan.ops)} eval={ms:.3f} ms", flush=True)

# the sampler path: time table + conditioning part of the init conv once per trajectory, plan body per eval
ctx = unet.begin_sampling(cond, torch.linspace(-3, 3, 8, device=dev))
for k in range(8):
    unet.eval_prepared(ctx, x, k % 8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(200):
    unet.eval_prepared(ctx, x, k % 8)
torch.cuda.synchronize()
print(f"B={B} sampler path: body ops={plan.n_body_ops} eval={(time.perf_counter() - t0) / 200 * 1e3:.3f} ms (incl. the time-row copy)", flush=True)
