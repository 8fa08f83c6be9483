import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_unet_ops import _run, _op, _pack_conv, bf, DEV
g = torch.Generator().manual_seed(3)
B, H, C = 2, 8, 128
x = torch.randn(B, C, H, H, generator=g)
w3 = torch.randn(C, C, 3, 3, generator=g) / 34; b3 = torch.randn(C, generator=g)
p3, _ = _pack_conv(w3)
xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
ref = F.conv2d(bf(x), bf(w3), b3, padding=1).permute(0, 2, 3, 1)
for a_f32 in (True, False):
    xin = xd if a_f32 else xd.to(torch.bfloat16)
    for tile in ((2, 4), (2, 2), (4, 4), (1, 4), (4, 2), (2, 1)):
        for groups in (1, 2):
            out = torch.zeros(B, H, H, C, device=DEV)
            _run([_op(1, 1 if a_f32 else 0, p=(xin, p3, b3.to(DEV), out, None), i=(B, H, H, C, H, H, C, C, 0, 3, 3, 1, 1, groups, tile[0] * 16 + tile[1]))])
            print(a_f32, tile, groups, 'maxerr', (out.cpu() - ref).abs().max().item())
