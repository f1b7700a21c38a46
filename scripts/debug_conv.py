import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from implementation_phd_lab_vision_amd import ops, _lib
_lib.build_library(); _lib.load_library()
case = (1, 56, 56, 64, 256, 1, 1, 0, True, True)
n, h, w, cin, cout, k, stride, pad, relu, has_res = case
g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
x = (torch.randn((n, cin, h, w), generator=g)).to(torch.bfloat16)
wt = (torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16)
bias = torch.randn(cout, generator=g) * 0.1
res = torch.randn((n, cout, h, w), generator=g).to(torch.bfloat16)
pre = F.conv2d(x.double(), wt.double()) + bias.double().view(1, -1, 1, 1) + res.double()
ref = F.relu(pre).float().to(torch.bfloat16).float()
d = torch.device('cuda:0')
xd = x.permute(0, 2, 3, 1).contiguous().to(d); wd = wt.permute(0, 2, 3, 1).contiguous().to(d)
bd = bias.to(d); rd = res.permute(0, 2, 3, 1).contiguous().to(d)
for tile in (0, 1, 2, 3, 4, 5, 17, 18, 19, 20, 21):
    for rep in range(2):
        y = ops.conv2d_bf16(xd, wd, bd, relu=True, residual=rd, tile=tile)
        torch.cuda.synchronize()
        got = y.float().cpu().permute(0, 3, 1, 2)
        diff = (got - ref).abs()
        bad = diff > ref.abs().clamp_min(2.0**-20) * 2.0**-7
        idx = bad.nonzero()
        print(f"tile {tile} rep {rep}: nbad={len(idx)} ndiff={(diff>0).sum().item()}")
        for i in idx[:6]:
            i = tuple(i.tolist())
            print("   ", i, "got", got[i].item(), "ref", ref[i].item(), "pre64", pre[i].item())
