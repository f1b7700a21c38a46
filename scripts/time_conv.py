"""Time conv shapes x tiles in one process.  usage: python scripts/time_conv.py "B,H,cin,cout,k,s,res;..." "tiles csv" [iters]"""
import sys, torch
sys.path.insert(0, '.')
from implementation_phd_lab_vision_amd import ops, _lib
_lib.load_library()
shapes = [tuple(int(v) for v in s.split(',')) for s in sys.argv[1].split(';')]
tiles = [int(t) for t in sys.argv[2].split(',')]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
d = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
for (B, H, cin, cout, k, s, res) in shapes:
    pad = 1 if k == 3 else 0
    ho = (H + 2 * pad - k) // s + 1
    x = torch.randn((B, H, H, cin), generator=g).to(torch.bfloat16).to(d)
    w = (torch.randn((cout, k, k, cin), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).to(d)
    bias = torch.randn(cout, generator=g).to(d)
    r = torch.randn((B, ho, ho, cout), generator=g).to(torch.bfloat16).to(d) if res else None
    flops = 2.0 * B * ho * ho * cout * cin * k * k
    out = []
    for tile in tiles:
        try:
            for _ in range(2): ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): ops.conv2d_bf16(x, w, bias, stride=s, pad=pad, relu=True, residual=r, tile=tile)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            out.append(f"{tile}:{us:.1f}us/{flops/us/1e6:.0f}TF")
        except Exception as e:
            out.append(f"{tile}:ERR")
    print(f"{H}x{H} {cin}->{cout} k{k} s{s} res{res}: " + ' '.join(out), flush=True)
