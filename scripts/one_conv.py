"""Run ONE conv shape with ONE tile variant N times (for rocprofv3 --pmc / --kernel-trace).
usage: python scripts/one_conv.py B H Cin Cout k stride res tile iters"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
B, H, cin, cout, k, s, res, tile, iters = [int(v) for v in sys.argv[1:10]]
d = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
pad = 1 if k == 3 else 0
ho = (H + 2 * pad - k) // s + 1
x = torch.randn((B, H, H, cin), generator=g).to(torch.bfloat16).to(d)
w = (torch.randn((cout, k, k, cin), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).to(d)
bias = torch.randn(cout, generator=g).to(d)
r = torch.randn((B, ho, ho, cout), generator=g).to(torch.bfloat16).to(d) if res else None
for _ in range(iters):
    y = ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
torch.cuda.synchronize()
print('done', float(y.float().abs().mean()))
